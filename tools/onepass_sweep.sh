#!/bin/bash
# tools/onepass_sweep.sh <tag>: the one-launch round (PBWTAMD_ONEPASS=1, skel_onepass_kernel) against the three-launch round: parity, then us/site by width,
# chain alone and with the bench consumers.  Output: gpurun_out/<tag>/
tag=${1:-r5d}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "onepass" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
fi
wb() { env "$@" timeout 200 python tools/wide_bench.py $M $S $W 2>&1 | tail -1; }
{
for M in ${WIDTHS:-100000 10000 30000 50000 250000 500000}; do
  S=16384
  for W in none hp; do
    echo "== M $M $W three launches"; wb PBWTAMD_ONEPASS=0; wb PBWTAMD_ONEPASS=0
    echo "== M $M $W one launch"; wb PBWTAMD_ONEPASS=1; wb PBWTAMD_ONEPASS=1
  done
done
} > $out/widths.txt 2>&1
cat $out/widths.txt
