#!/bin/bash
# round 4, call c: XCC_ID team probe; kernel stats of the two fill forms at 1 M and 100 k; posshard tests after the sticky-error change
out=gpurun_out/r4c; mkdir -p $out
timeout 200 ./tools/latprobe4 > $out/latprobe4.txt 2>&1; echo "latprobe4 rc=$?"; grep "team on" $out/latprobe4.txt | head -40
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for seq in 1 0; do
  for M in 1000000 100000; do
    PBWTAMD_FILL_SEQ=$seq timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_${seq}_$M -o w -- python tools/wide_bench.py $M 4096 hp > $out/tr_${seq}_$M.log 2>&1
    echo "FILL_SEQ=$seq M=$M: $(tail -1 $out/tr_${seq}_$M.log)"
    f=$(find $out/tr_${seq}_$M -name "*kernel_stats.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  done
done 2>&1 | tee $out/fill_ab.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "position_sharded_chain" > $out/pytest_multi.log 2>&1; tail -5 $out/pytest_multi.log
