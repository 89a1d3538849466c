#!/bin/bash
# tools/opts_1m.sh <tag>: the north-star width by consumer set — chain only, + maxWithin histogram (h), + pack3 (p), both (hp) — us/site end to end and per chain launch
tag=${1:-r5s}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ for W in none p h hp; do for i in 1 2; do timeout 300 python tools/wide_bench.py ${M:-1000000} ${SITES:-8192} $W 2>&1 | tail -1; done; done; } > $out/opts.txt 2>&1; cat $out/opts.txt
