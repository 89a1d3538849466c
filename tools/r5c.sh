#!/bin/bash
out=gpurun_out/r5c; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or (packed_fill_every_position and seq) or long_walks or without_ids" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench.json").read().strip().split("\n")[-1])
print("value %.3e ms/step %.3f frac %.4f us/launch %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_launch"]))
ns = d.get("north_star_width", {}); print("north-star us/site %.3f frac %.3f hist %s" % (ns.get("us_per_site", 0), ns.get("whole_job_frac_of_hbm_peak", 0), ns.get("within_reports_hist_total")))
mp = d.get("many_panels", {}); print("many_panels %.3e x%.2f" % (mp.get("value", 0), mp.get("speedup_vs_one_panel", 0)))
md = d.get("match_dynamic", {}); print("match_dynamic us/site %.2f records %s" % (md.get("us_per_site", 0), md.get("records")))
print("cpu_baseline", d.get("cpu_baseline"))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $out/sq -o wide -- python tools/wide_bench.py 1000000 2048 hp > $out/sq.log 2>&1
python - <<'PY'
import csv, collections, glob
p = glob.glob("gpurun_out/r5c/sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(p)):
    d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "SQ_WAVES" not in c or not ("sweep_hist" in k or "fillseq" in k or "skel_" in k): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = max(m["SQ_WAVES"], 1)
    print("%-62s waves %8d per wave: VALU %6d SALU %6d LDS %5d VMEM_WR %5d cyc %7d wait %7d (%2d%%) active %6d" % (k[:62], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m.get("SQ_INSTS_VMEM_WR", 0) / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1), m["SQ_ACTIVE_INST_ANY"] / w))
PY
rm -rf $out/sq
