mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python tools/wide_bench.py 1000000 8192 hp; env "$@" timeout 300 python tools/wide_bench.py 1000000 8192 none; }
run PBWTAMD_K2_TPW=32
run PBWTAMD_K2_TPW=64
run PBWTAMD_K2_TPW=6432
run PBWTAMD_K2_TPW=32
