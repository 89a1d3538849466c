#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
# (1) multi-rank tests incl. world 4 and 8 on one GPU
timeout 1800 python -m pytest tests/test_gpu_multi.py -x -q > $out/pytest_multi.log 2>&1; tail -6 $out/pytest_multi.log
# (2) bench.py at N = 2 (gloo, both ranks on this GPU): replicas + the attached position_sharded object
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --ps-sites 4096 > $out/bench_n2.json 2> $out/bench_n2.err
tail -3 $out/bench_n2.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4j/bench_n2.json").read().strip().split("\n")[-1])
    print("N=2 value %.3e scaling %s; position_sharded: %s" % (d["value"], d["scaling"], json.dumps({k: v for k, v in d.get("position_sharded", {}).items() if k not in ("roofline", "exchange")})))
except Exception as ex:
    print("N=2 line not parsed:", ex)
PY
# (3) one rank in shard mode at 1 M (the launch structure a rank of a multi-GPU job runs)
timeout 600 python bench.py --mode posshard --backend gloo --haps 1000000 --steps 1 --warmup 1 > $out/posshard_1rank.json 2> $out/posshard_1rank.err; python -c "
import json; d=json.loads(open('$out/posshard_1rank.json').read().strip().split('\n')[-1]); print('one rank in shard mode: %.2f us/site, %.2f us/launch' % (d['ms_per_step']*1e3/8192, d['roofline']['us_per_launch']))"
# (4) configs[4] at its own length: 1 M haplotypes x 10 M sites, panel generated per step (no pack3: the bytes of 10 M sites do not fit), then the same with 1 M sites incl. pack3
timeout 900 python bench.py --stream-panel --ns-sites 10000000 --ns-no-pack3 > $out/c5_full.json 2> $out/c5_full.err; tail -2 $out/c5_full.err; cat $out/c5_full.json | head -c 1500; echo
timeout 600 python bench.py --stream-panel --ns-sites 1000000 > $out/c5_1m_streamed.json 2> $out/c5_1m.err; cat $out/c5_1m_streamed.json | head -c 1200; echo
