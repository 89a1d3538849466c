#!/bin/bash
out=gpurun_out/r4u; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or packed_fill_every_position or without_ids or long_walks or many_panels" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() { echo "$1: $(env $2 timeout 200 python tools/wide_bench.py $3 $4 hp 2>&1 | tail -1)"; }
{ for i in 1 2; do
  run "p16  1M  " "X=1" 1000000 8192
  run "p32  1M  " "PBWTAMD_P16=0" 1000000 8192
  run "p16  100k" "X=1" 100000 16384
  run "p32  100k" "PBWTAMD_P16=0" 100000 16384
done
  run "p16  100k iid" "KIND=1" 100000 8192
  run "p32  100k iid" "KIND=1 PBWTAMD_P16=0" 100000 8192
} > $out/ab.txt 2>&1
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1000000 100000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$M -o w -- python tools/wide_bench.py $M 4096 hp > $out/tr_$M.log 2>&1
  f=$(find $out/tr_$M -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out/tr_$M
done 2>&1 | tee $out/stats.txt
