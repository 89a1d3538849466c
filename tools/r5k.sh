#!/bin/bash
out=gpurun_out/r5k; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or packed_fill_every_position or without_ids or long_walks or golden or merge1" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
bash tools/ab.sh $out/ab_1m.txt 1000000 8192 2 "p16=X=1"
bash tools/ab.sh $out/ab_100k.txt 100000 131072 2 "p16=X=1"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1000000 100000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$M -o w -- python tools/wide_bench.py $M 4096 hp > $out/tr_$M.log 2>&1
  f=$(find $out/tr_$M -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out/tr_$M
done 2>&1 | tee $out/stats.txt
