mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 100000 200000 250000 400000 600000 1000000; do timeout 300 python tools/wide_bench.py $M 16384 hp; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
