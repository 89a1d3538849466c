mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
echo "100k"; timeout 300 python tools/wide_bench.py 100000 16384 hp; PBWTAMD_SKT=1024 timeout 300 python tools/wide_bench.py 100000 16384 hp
echo "200k"; timeout 300 python tools/wide_bench.py 200000 8192 hp; PBWTAMD_SKT=1024 timeout 300 python tools/wide_bench.py 200000 8192 hp
