#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
PBWTAMD_PACKED_UC=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_fill_every_position and seq" > $out/pytest_uc.log 2>&1; tail -3 $out/pytest_uc.log
{ for v in 0 1; do
    echo "PACKED_UC=$v 1M: $(PBWTAMD_PACKED_UC=$v timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1)"
    echo "PACKED_UC=$v 100k: $(PBWTAMD_PACKED_UC=$v timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1)"
  done
} > $out/ab.txt 2>&1
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PBWTAMD_PACKED_UC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o w -- python tools/wide_bench.py 1000000 4096 hp > $out/tr.log 2>&1
f=$(find $out/tr -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/tr
