#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "packed_fill_every_position or without_ids or both_chains" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
{ echo "shipped lib"; for M in 1000000 100000; do timeout 200 python tools/wide_bench.py $M 8192 hp 2>&1 | tail -1; done
  export PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_measure.so
  for v in "X=1" "PBWTAMD_S2_CUS=224" "PBWTAMD_S2_CUS=192" "PBWTAMD_FILL_PAD_KB=5" "PBWTAMD_FILL_PAD_KB=9" "PBWTAMD_NOFILL=1" "PBWTAMD_DEBUG_FILL_NOWRITE=1"; do
    echo "measure lib, $v 1M"; env $v timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1
  done
  echo "chain alone"; timeout 200 python tools/wide_bench.py 1000000 8192 none 2>&1 | tail -1
  for v in "X=1" "PBWTAMD_S2_CUS=128" "PBWTAMD_S2_CUS=192" "PBWTAMD_S2_CUS=0"; do
    echo "measure lib, $v 100k"; env $v timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1
  done
  echo "chain alone 100k"; timeout 200 python tools/wide_bench.py 100000 16384 none 2>&1 | tail -1
} > $out/ab.txt 2>&1
cat $out/ab.txt
unset PBWTAMD_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1000000 100000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$M -o w -- python tools/wide_bench.py $M 4096 hp > $out/tr_$M.log 2>&1
  echo "M=$M: $(tail -1 $out/tr_$M.log)"
  f=$(find $out/tr_$M -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("   %-60s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out/tr_$M
done 2>&1 | tee $out/stats.txt
