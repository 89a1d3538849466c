#!/bin/bash
# tools/xcd_ab.sh <tag>: the one-launch round with tiles dealt to the XCDs in contiguous ranges (PBWTAMD_XCD=7) against tile = workgroup index (PBWTAMD_XCD=5: a tile
# then waits only for workgroups dispatched before it); measurement build
tag=${1:-r5u}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/pbwt_amd/libpbwtgpu_measure.so
{ for M in ${WIDTHS:-100000 30000 150000}; do for X in 7 5; do for W in none hp; do for i in 1 2; do
  echo -n "XCD=$X "; PBWTAMD_LIB=$L PBWTAMD_XCD=$X timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; done; } > $out/xcd.txt 2>&1; cat $out/xcd.txt
