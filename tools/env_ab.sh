#!/bin/bash
# tools/env_ab.sh <tag> <VAR> <value> [<value> ...]: us/site of the chain alone ('none') and beside the bench consumers ('hp') with VAR set to each value in turn,
# interleaved; TESTS="<pytest -k expression>": the chain's parity tests first, with VAR at its last value; PROF=1: the one-launch round's stamps at the last value
tag=$1; var=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
last=${@: -1}
if [ -n "$TESTS" ]; then env $var=$last timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$TESTS" > $out/tests.txt 2>&1; tail -3 $out/tests.txt; fi
{ for M in ${WIDTHS:-100000 30000 150000}; do for W in ${OPTS:-none hp}; do for i in $(seq ${REPS:-2}); do for v in "$@"; do
  echo -n "$var=$v "; env $var=$v timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; done; } > $out/ab.txt 2>&1; cat $out/ab.txt
if [ -n "$PROF" ]; then env $var=$last PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py 100000 4096 none > $out/tiles.txt 2>&1; grep "onepass prof" $out/tiles.txt; fi
