#!/bin/bash
# tools/lb_variants.sh <tag> <lib suffixes...>: builds of the look-back-wave round (pbwt_amd/libpbwtgpu_<suffix>.so; "" = the shipped library), us/site and stamps
tag=${1:-r5lv}; shift; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  L=$PWD/pbwt_amd/libpbwtgpu_$v.so; [ "$v" = base ] && L=$PWD/pbwt_amd/libpbwtgpu.so
  for W in none hp; do for i in 1 2; do echo -n "$v "; PBWTAMD_LIB=$L PBWTAMD_ONEPASS_LB=${LB:-1} timeout 200 python tools/wide_bench.py ${M:-100000} 16384 $W 2>&1 | tail -1; done; done
  PBWTAMD_LIB=$L PBWTAMD_ONEPASS_LB=${LB:-1} PBWTAMD_ONEPASS_PROF=2 timeout 200 python tools/wide_bench.py ${M:-100000} 4096 none > $out/tiles_$v.txt 2>&1; grep "onepass prof" $out/tiles_$v.txt
done > $out/variants.txt 2>&1
cat $out/variants.txt
