#!/bin/bash
# tools/tile_ab.sh <tag>: 256- against 512-position tiles under the one-launch round (measurement build: PBWTAMD_SKT), us/site with the bench consumers
tag=${1:-r5r}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/pbwt_amd/libpbwtgpu_measure.so
{ for M in ${WIDTHS:-10000 12000 60000 70000 80000 100000}; do for T in 256 512; do for W in none hp; do
  echo -n "T=$T "; PBWTAMD_LIB=$L PBWTAMD_SKT=$T PBWTAMD_ONEPASS_MAXW=512 timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; } > $out/tiles.txt 2>&1; cat $out/tiles.txt
