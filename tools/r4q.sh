#!/bin/bash
out=gpurun_out/r4q; mkdir -p $out
run() { echo "$1: $(env $2 timeout 200 python tools/wide_bench.py $3 $4 hp 2>&1 | tail -1)"; }
{ for i in 1 2; do
  run "nt stores    1M  " "X=1" 1000000 8192
  run "plain stores 1M  " "PBWTAMD_LIB=$GRAFT_REPO_ROOT/pbwt_amd/libpbwtgpu_measure.so" 1000000 8192
  run "nt stores    100k" "X=1" 100000 16384
  run "plain stores 100k" "PBWTAMD_LIB=$GRAFT_REPO_ROOT/pbwt_amd/libpbwtgpu_measure.so" 100000 16384
done; } > $out/ab.txt 2>&1
cat $out/ab.txt
