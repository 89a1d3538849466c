#!/bin/bash
# tools/lb_ab.sh <tag>: the one-launch round with look-back waves (PBWTAMD_ONEPASS_LB=1) against the shipped form (=0): parity of the chain tests, then us/site alone
# and beside the bench consumers, then the per-tile stamps
tag=${1:-r5lb}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
PBWTAMD_ONEPASS_LB=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onepass" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
fi
{ for M in ${WIDTHS:-100000 30000 150000}; do for W in none hp; do for LB in 0 1; do for i in 1 2; do
  echo -n "LB=$LB "; PBWTAMD_ONEPASS_LB=$LB timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; done; } > $out/ab.txt 2>&1; cat $out/ab.txt
PBWTAMD_ONEPASS_LB=1 PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py 100000 4096 none > $out/tiles_lb.txt 2>&1; grep "onepass prof" $out/tiles_lb.txt
