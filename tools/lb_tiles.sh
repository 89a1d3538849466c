#!/bin/bash
# tools/lb_tiles.sh <tag>: per-tile stamps of the one-launch round with look-back waves
tag=${1:-r5lt}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in ${WIDTHS:-100000}; do for W in none; do for i in 1 2; do
  PBWTAMD_ONEPASS_LB=1 PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py $M 4096 $W > $out/tiles_${M}_${W}_$i.txt 2>&1
done; done; done
grep "us/site\|onepass prof" $out/tiles_*_1.txt
