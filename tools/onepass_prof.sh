#!/bin/bash
tag=${1:-r5e}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
for M in ${WIDTHS:-100000 30000 250000}; do for W in none hp; do
  echo "== M $M $W"; PBWTAMD_ONEPASS=1 PBWTAMD_ONEPASS_PROF=1 timeout 200 python tools/wide_bench.py $M 4096 $W 2>&1 | tail -7
done; done
} > $out/prof.txt 2>&1
cat $out/prof.txt
