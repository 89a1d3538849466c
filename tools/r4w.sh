#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or (packed_fill_every_position and seq) or long_walks" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() { echo "$1: $(env $2 timeout 200 python tools/wide_bench.py $3 $4 hp 2>&1 | tail -1)"; }
{ for i in 1 2; do
  for it in 8 4 2; do
  run "p16 iters $it 1M  " "PBWTAMD_P16_ITERS=$it" 1000000 8192
  run "p16 iters $it 100k" "PBWTAMD_P16_ITERS=$it" 100000 16384
  done
  run "p32  1M  " "PBWTAMD_P16=0" 1000000 8192
  run "p32  100k" "PBWTAMD_P16=0" 100000 16384
done
  run "p16 iters 8 100k iid" "KIND=1 PBWTAMD_P16_ITERS=8" 100000 8192
  run "p16 iters 4 100k iid" "KIND=1 PBWTAMD_P16_ITERS=4" 100000 8192
  run "p32  100k iid" "KIND=1 PBWTAMD_P16=0" 100000 8192
} > $out/ab.txt 2>&1
cat $out/ab.txt
