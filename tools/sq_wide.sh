#!/bin/bash
# tools/sq_wide.sh <out> <M> <sites>: SQ counters per wave of the consumers and the chain on the bench path
out=$1; M=$2; sites=$3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $out/sq -o wide -- python tools/wide_bench.py $M $sites hp > $out/sq.log 2>&1
python - $out <<'PY'
import csv, collections, glob, sys
p = glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(p)):
    d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "SQ_WAVES" not in c or not ("sweep_hist" in k or "fillseq" in k): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = max(m["SQ_WAVES"], 1)
    print("%-62s waves %8d per wave: VALU %6d SALU %6d LDS %5d VMEM_WR %5d cyc %7d wait %7d (%2d%%) active %6d" % (k[:62], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m.get("SQ_INSTS_VMEM_WR", 0) / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1), m["SQ_ACTIVE_INST_ANY"] / w))
PY
rm -rf $out/sq
