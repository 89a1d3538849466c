#!/bin/bash
# why is the 16-bit sweep slower than the 32-bit one?  kernel times with the walks off (DEBUG_SWEEP=2: first step only), SQ counters of both
out=gpurun_out/r4s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PBWTAMD_LIB=$GRAFT_REPO_ROOT/pbwt_amd/libpbwtgpu_measure.so
stats() { f=$(find $out/tr -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    if "sweep_hist" in r["Name"] or "fillseq" in r["Name"]:
        print("   %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/tr; }
for v in "PBWTAMD_P16=1" "PBWTAMD_P16=0" "PBWTAMD_P16=1 PBWTAMD_DEBUG_SWEEP=2" "PBWTAMD_P16=0 PBWTAMD_DEBUG_SWEEP=2" "PBWTAMD_P16=1 PBWTAMD_DEBUG_SWEEP=1" ; do
  echo "== $v"; env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o w -- python tools/wide_bench.py 1000000 2048 hp > $out/tr.log 2>&1; tail -1 $out/tr.log | cut -c1-150; stats
done 2>&1 | tee $out/stats.txt
for v in "PBWTAMD_P16=1" "PBWTAMD_P16=0"; do
env $v timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $out/sq -o wide -- python tools/wide_bench.py 1000000 2048 hp > $out/sq.log 2>&1
echo "== $v"; python - <<'PY'
import csv, collections, glob
p = glob.glob("gpurun_out/r4s/sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(p)):
    d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "SQ_WAVES" not in c or not ("sweep_hist" in k or "fillseq" in k): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = max(m["SQ_WAVES"], 1)
    print("%-62s waves %8d per wave: VALU %6d SALU %6d LDS %5d VMEM_RD %5d cyc %7d wait %7d (%2d%%) active %6d" % (k[:62], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m.get("SQ_INSTS_VMEM_RD", 0) / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1), m["SQ_ACTIVE_INST_ANY"] / w))
PY
rm -rf $out/sq
done 2>&1 | tee $out/sq.txt
