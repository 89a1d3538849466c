cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; for M in 10000 100000 250000 1000000; do for O in h hp; do env "$@" timeout 300 python tools/wide_bench.py $M 16384 $O; done; done; }
run PBWTAMD_ASYNC_FLUSH=0
run PBWTAMD_ASYNC_FLUSH=1
run PBWTAMD_ASYNC_FLUSH=1 PBWTAMD_FLUSH_POST_FIRST=0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
