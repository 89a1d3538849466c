#!/bin/bash
out=gpurun_out/r5f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/md_bench.py 16384 2 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o md -- python tools/md_bench.py 8192 1 > $out/tr.log 2>&1
f=$(find $out/tr -name "*kernel_stats.csv" | head -1); cp $f $out/md_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("   %-66s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("total kernel time %.1f ms" % (tot / 1e6))
PY
python tools/trace_busy.py $(find $out/tr -name "*kernel_trace.csv" | head -1) 2>/dev/null | tail -12
rm -rf $out/tr
