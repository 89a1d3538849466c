#!/bin/bash
# tools/full_gpu.sh <tag>: the whole -m gpu suite, then the driver's bench command; output in gpurun_out/<tag>/
tag=${1:-r5j}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; tail -5 $out/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value %.3e frac %.3f us/launch %.2f" % (d["value"], d["roofline"]["frac"], d["roofline"]["us_per_launch"]))
for k in ("many_panels", "north_star_width", "match_dynamic", "cpu_baseline"):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if kk in ("value", "us_per_site", "panels", "tried", "whole_job_frac_of_hbm_peak")})
PY
