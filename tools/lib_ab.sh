#!/bin/bash
# tools/lib_ab.sh <tag> <other lib under pbwt_amd/>: the shipped library against another build, interleaved, us/site at a few widths (chain alone / with the bench consumers)
tag=${1:-r5ab}; other=$2; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ for M in ${WIDTHS:-100000 30000 150000}; do for W in none hp; do for i in 1 2; do
  echo -n "A "; timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1
  echo -n "B "; PBWTAMD_LIB=$PWD/pbwt_amd/$other timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1
done; done; done; } > $out/ab.txt 2>&1; cat $out/ab.txt
if [ -n "$TESTS" ]; then PBWTAMD_LIB=$PWD/pbwt_amd/$other timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$TESTS" 2>&1 | tail -2; fi
