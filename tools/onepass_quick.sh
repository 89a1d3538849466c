#!/bin/bash
# tools/onepass_quick.sh <tag> [lib]: stamps + us/site of the one-launch round at a few widths (A/B of kernel variants; PBWTAMD_LIB selects the build)
tag=${1:-r5f}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wb() { env "$@" timeout 200 python tools/wide_bench.py $M $S $W 2>&1 | tail -${TL:-1}; }
{
for M in ${WIDTHS:-100000 30000}; do
  S=4096; W=none; TL=6; echo "== prof M $M"; wb PBWTAMD_ONEPASS=1 PBWTAMD_ONEPASS_PROF=1
  S=16384; TL=1
  for W in none hp; do
    echo "== M $M $W three launches"; wb PBWTAMD_ONEPASS=0
    echo "== M $M $W one launch"; wb PBWTAMD_ONEPASS=1; wb PBWTAMD_ONEPASS=1
    for L in $LIBS; do echo "== M $M $W one launch, $L"; wb PBWTAMD_ONEPASS=1 PBWTAMD_LIB=$PWD/pbwt_amd/$L; wb PBWTAMD_ONEPASS=1 PBWTAMD_LIB=$PWD/pbwt_amd/$L; done
  done
done
} > $out/quick.txt 2>&1
cat $out/quick.txt
