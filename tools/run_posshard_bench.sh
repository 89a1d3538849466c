#!/bin/bash
# tools/run_posshard_bench.sh WORLD M STEPS [extra bench args] — bench.py --mode posshard with WORLD ranks sharing this box's GPU (gloo)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=$1; M=$2; K=$3; shift 3
mkdir -p gpurun_out/r3
if [ "$W" = 1 ]; then timeout ${PS_TIMEOUT:-600} python bench.py --mode posshard --backend gloo --gpus 1 --haps $M --steps $K --warmup 1 "$@" > gpurun_out/r3/psbench_${W}_${M}.json 2> gpurun_out/r3/psbench_${W}_${M}.err
else timeout ${PS_TIMEOUT:-600} python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((20000 + RANDOM % 20000)) bench.py --mode posshard --backend gloo --gpus $W --haps $M --steps $K --warmup 1 "$@" > gpurun_out/r3/psbench_${W}_${M}.json 2> gpurun_out/r3/psbench_${W}_${M}.err; fi
echo "rc=$? world=$W M=$M"; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/r3/psbench_${W}_${M}.err | tail -5 | cut -c1-300
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3/psbench_${W}_${M}.json") if l.startswith("{")][-1])
    print("value %.3e  us/site %.3f  chain us/launch %.2f  hist_total %d" % (d["value"], 1e3 * d["ms_per_step"] / d["config"]["sites_per_step"], d["roofline"]["us_per_launch"], d["within_reports_hist_total"]))
except Exception as e:
    print("no json:", e)
PY
