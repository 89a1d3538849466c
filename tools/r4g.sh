#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
{ for v in 0 256 64 16 4; do
    echo "FILL_WB=$v 1M: $(PBWTAMD_FILL_WB=$v timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1)"
  done
  for v in 0 64 8; do
    echo "FILL_WB=$v 100k: $(PBWTAMD_FILL_WB=$v timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1)"
  done
} > $out/ab.txt 2>&1
cat $out/ab.txt
