cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_measure.so
for i in 1 2 3; do for cfg in X=1 PBWTAMD_THR_DEPTH=8 PBWTAMD_FLUSH_AT=48 PBWTAMD_SWEEP_ITERS=8 "PBWTAMD_THR_DEPTH=8 PBWTAMD_FLUSH_AT=48"; do echo -n "$cfg  "; env $cfg timeout 300 python tools/wide_bench.py 1000000 16384 hp 2>&1 | tail -1 | cut -c28-56; done; done
