#!/bin/bash
out=gpurun_out/r5e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or (packed_fill_every_position and seq) or without_ids" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
bash tools/ab.sh $out/ab_1m.txt 1000000 8192 2 "p16=X=1"
bash tools/ab.sh $out/ab_100k.txt 100000 131072 2 "p16=X=1"
run() { timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
for P in 1 2 3 4 6 8; do run --panels $P; done 2>&1 | tee $out/panels.txt
