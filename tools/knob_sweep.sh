#!/bin/bash
# orchestration knobs of a measurement build at one width: us/site end to end of tools/wide_bench.py (usage: knob_sweep.sh M sites)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_measure.so
M=${1:-100000}; S=${2:-32768}
run() { echo -n "$1  "; env $1 timeout 300 python tools/wide_bench.py $M $S hp 2>&1 | tail -1 | cut -c28-110; }
run X=1; run X=1
for v in 0 16 32 48 60; do run PBWTAMD_FLUSH_AT=$v; done
for v in 1 3 4 8; do run PBWTAMD_THR_DEPTH=$v; done
for v in 8 16 56 0; do run PBWTAMD_THR_ROUNDS=$v; done
for v in 1 2 4 8; do run PBWTAMD_SWEEP_ITERS=$v; done
for v in 0 1 3 7; do run PBWTAMD_XCD=$v; done
run X=1
