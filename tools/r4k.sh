#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "synth_generator or build_and_within or both_chains" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() { timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
{ for P in 1 2 3 4; do run --panels $P; done; } 2>&1 | tee $out/panels.txt
timeout 600 python bench.py --stream-panel --ns-sites 1000000 > $out/c5_1m_streamed.json 2> $out/c5_1m.err; cat $out/c5_1m_streamed.json | head -c 600; echo
timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1
