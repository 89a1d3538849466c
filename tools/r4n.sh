#!/bin/bash
# fill: uniform-level fast path.  Parity on the fill's tests, then the two widths.
out=gpurun_out/r4n; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_fill_every_position or without_ids or both_chains or many_panels" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{ echo "1M: $(timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1)"
  echo "1M: $(timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1)"
  echo "100k: $(timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1)"
  echo "100k iid: $(KIND=1 timeout 200 python tools/wide_bench.py 100000 8192 hp 2>&1 | tail -1)"
} > $out/ab.txt 2>&1
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1000000 100000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$M -o w -- python tools/wide_bench.py $M 4096 hp > $out/tr_$M.log 2>&1
  echo "M=$M: $(tail -1 $out/tr_$M.log)"
  f=$(find $out/tr_$M -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out/tr_$M
done 2>&1 | tee $out/stats.txt
