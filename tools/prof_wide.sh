mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python tools/wide_bench.py 1000000 8192 hp; }
for V in 0 1 2 3 4 0; do run PBWTAMD_K2_VAR=$V; done
for I in 2 4 16 32; do run PBWTAMD_SWEEPH_ITERS=$I; done
for V in 0 1 2 4; do echo "alone $V"; PBWTAMD_K2_VAR=$V timeout 300 python tools/wide_bench.py 1000000 8192 none; done
