#!/bin/bash
# tools/scan_ab.sh <tag>: the scanner form of the one-launch round against three launches per round, widths above 320 tiles: us/site with the bench consumers (hp) and the chain alone (none)
tag=${1:-r6scan}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ for M in ${WIDTHS:-200000 300000 500000 1000000}; do for S in 0 1; do for W in none hp; do for i in 1 2; do
  echo -n "SCAN=$S "; PBWTAMD_ONEPASS_SCAN=$S timeout 300 python tools/wide_bench.py $M ${SITES:-8192} $W 2>&1 | tail -1; done; done; done; done; } > $out/ab.txt 2>&1; cat $out/ab.txt
