#!/bin/bash
# round 4, call b: latprobe4 (bounded), the packed fill at every position on both forms, A/B timing of the two fills
out=gpurun_out/r4b; mkdir -p $out
timeout 150 ./tools/latprobe4 > $out/latprobe4.txt 2>&1; echo "latprobe4 rc=$?"; tail -3 $out/latprobe4.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_fill_every_position" > $out/pytest_fill.log 2>&1; tail -15 $out/pytest_fill.log
for seq in 1 0; do
  echo "FILL_SEQ=$seq 1M"; PBWTAMD_FILL_SEQ=$seq timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1
  echo "FILL_SEQ=$seq 100k"; PBWTAMD_FILL_SEQ=$seq timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1
done > $out/ab.txt 2>&1
cat $out/ab.txt
