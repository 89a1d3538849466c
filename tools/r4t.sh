#!/bin/bash
out=gpurun_out/r4t; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "p16 or (packed_fill_every_position and seq)" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PBWTAMD_LIB=$GRAFT_REPO_ROOT/pbwt_amd/libpbwtgpu_measure.so
stats() { f=$(find $out/tr -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    if "sweep_hist_kernel<true" in r["Name"] or "fillseq" in r["Name"]:
        print("   %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/tr; }
for M in 1000000 100000; do
for v in "PBWTAMD_P16=1" "PBWTAMD_P16=0" "PBWTAMD_P16=1 PBWTAMD_DEBUG_SWEEP=3" "PBWTAMD_P16=0 PBWTAMD_DEBUG_SWEEP=3" "PBWTAMD_P16=0 PBWTAMD_DEBUG_SWEEP=2"; do
  echo "== M=$M $v"; env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o w -- python tools/wide_bench.py $M 2048 hp > $out/tr.log 2>&1; tail -1 $out/tr.log | cut -c1-150; stats
done; done 2>&1 | grep -v "tool finalization" | tee $out/stats.txt
