#!/bin/bash
# tools/onepass_tiles.sh <tag>: the one-launch round's stamps of EVERY tile of the last launch (PBWTAMD_ONEPASS_PROF=2), chain alone and beside the bench consumers
tag=${1:-r5t}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in ${WIDTHS:-100000}; do for W in none hp; do for i in 1 2; do
  PBWTAMD_ONEPASS=1 PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py $M 4096 $W > $out/tiles_${M}_${W}_$i.txt 2>&1
done; done; done
for f in $out/tiles_*_1.txt; do grep "us/site\|onepass prof" $f; done
