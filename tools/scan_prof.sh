#!/bin/bash
# tools/scan_prof.sh <tag>: the scanner form's timeline at the north-star width (every tile's and scanner's stamps of the last launch)
tag=${1:-r6scanprof}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in ${WIDTHS:-1000000}; do for W in ${OPTS:-none hp}; do
  PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py $M 2048 $W > $out/prof_${M}_$W.txt 2>&1; grep -v "onepass tile\|onepass scanner" $out/prof_${M}_$W.txt
done; done
