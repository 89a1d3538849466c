#!/bin/bash
# what bounds the consumers at 1 M: SQ counters of the new fill; fill without stores; sweep without walks / atomics; consumers priced one by one
out=gpurun_out/r4o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WIDE="python tools/wide_bench.py 1000000 2048 hp"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/wide_sq -o wide -- $WIDE > $out/wide_sq.log 2>&1
python - <<'PY'
import csv, collections, glob
p = glob.glob("gpurun_out/r4o/wide_sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(p)):
    d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "SQ_WAVES" not in c: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = max(m["SQ_WAVES"], 1)
    print("%-62s waves %8d per wave: VALU %6d SALU %6d LDS %5d cyc %7d wait %7d (%2d%%) active %6d" % (k[:62], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1), m["SQ_ACTIVE_INST_ANY"] / w))
PY
rm -rf $out/wide_sq
export PBWTAMD_LIB=$GRAFT_REPO_ROOT/pbwt_amd/libpbwtgpu_measure.so
run() { echo "$1: $(env $2 timeout 200 python tools/wide_bench.py 1000000 8192 ${3:-hp} 2>&1 | tail -1)"; }
{ run "shipped            " "X=1"
  run "fill no stores     " "PBWTAMD_DEBUG_FILL_NOWRITE=1"
  run "no fill            " "PBWTAMD_NOFILL=1"
  run "sweep first step   " "PBWTAMD_DEBUG_SWEEP=2"
  run "sweep no atomics   " "PBWTAMD_DEBUG_SWEEP=1"
  run "hist only (no p3)  " "X=1" h
  run "pack3 only         " "X=1" p
  run "chain only         " "X=1" none
} > $out/pricing.txt 2>&1
cat $out/pricing.txt
for v in "X=1" "PBWTAMD_DEBUG_FILL_NOWRITE=1"; do
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o w -- python tools/wide_bench.py 1000000 4096 hp > $out/tr.log 2>&1
  f=$(find $out/tr -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out/tr
done 2>&1 | tee $out/stats.txt
