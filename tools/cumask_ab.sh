#!/bin/bash
# tools/cumask_ab.sh <tag>: CUs the consumer stream is confined to (measurement build, PBWTAMD_S2_CUS) under the one-launch round, us/site with the bench consumers
tag=${1:-r5cu}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/pbwt_amd/libpbwtgpu_measure.so
{ for M in ${WIDTHS:-100000 30000 150000}; do for C in 64 96 128 160 192 0; do for i in 1 2; do
  echo -n "CUS=$C "; PBWTAMD_LIB=$L PBWTAMD_S2_CUS=$C timeout 200 python tools/wide_bench.py $M 16384 hp 2>&1 | tail -1; done; done; done; } > $out/cus.txt 2>&1; cat $out/cus.txt
