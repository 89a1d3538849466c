#!/bin/bash
out=gpurun_out/r5m; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_z_configs.py -x -q -k "both_chains or every_site or wider or read_side or without_ids or configs or p16" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
echo "1M chain only: $(timeout 200 python tools/wide_bench.py 1000000 8192 none 2>&1 | tail -1)"
echo "1M chain only: $(timeout 200 python tools/wide_bench.py 1000000 8192 none 2>&1 | tail -1)"
bash tools/ab.sh $out/ab_1m.txt 1000000 8192 2 "p16=X=1"
echo "2M: $(timeout 200 python tools/wide_bench.py 2000000 4096 hp 2>&1 | tail -1)"
echo "600k: $(timeout 200 python tools/wide_bench.py 600000 8192 hp 2>&1 | tail -1)"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o w -- python tools/wide_bench.py 1000000 4096 none > $out/tr.log 2>&1
f=$(find $out/tr -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:5]:
    print("   %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/tr
