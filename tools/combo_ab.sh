#!/bin/bash
# tools/combo_ab.sh <tag> "<ENV=.. ENV=..>" ...: us/site alone / beside the bench consumers for each environment set, interleaved (WIDTHS, REPS)
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ for M in ${WIDTHS:-100000}; do for W in ${OPTS:-none hp}; do for i in $(seq ${REPS:-2}); do for v in "$@"; do
  echo -n "[$v] "; env $v timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; done; } > $out/ab.txt 2>&1; cat $out/ab.txt
