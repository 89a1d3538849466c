#!/bin/bash
# tools/profile_round.sh <tag> — rocprofv3 evidence for one round, written under gpurun_out/<tag>/
# (kernel-trace stats and the PMC passes are separate runs, as the pool requires)
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu --no-1m"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $BENCH > $out/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- $BENCH > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- $BENCH > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o bench -- $BENCH > $out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calib_fetch -o calib -- ./tools/pmc_calib > $out/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/calib_write -o calib -- ./tools/pmc_calib > $out/calib_write.log 2>&1
# the north-star width (1 M haplotypes), same option set: kernel stats, traffic and SQ counters of chain + consumers
WIDE="python tools/wide_bench.py 1000000 2048 hp"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wide_trace -o wide -- $WIDE > $out/wide_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/wide_fetch -o wide -- $WIDE > $out/wide_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/wide_write -o wide -- $WIDE > $out/wide_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/wide_sq -o wide -- $WIDE > $out/wide_sq.log 2>&1
# -matchDynamic, 10 000 queries against 1 M haplotypes
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/qs_trace -o qs -- python tools/qsweep_bench.py 1000000 10000 4096 > $out/qs_trace.log 2>&1
# the position-sharded chain, one rank (the launch structure a rank of a multi-GPU job runs; the exchange degenerates)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_trace -o shard -- python bench.py --mode posshard --backend gloo --haps 1000000 --steps 1 --warmup 1 > $out/shard_trace.log 2>&1
# what the chain's launches cost beside each consumer kernel (north-star width): durations by the kernel running on the other stream
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/wide_tl -o wide -- python tools/wide_bench.py 1000000 4096 hp > $out/wide_tl.log 2>&1
{ grep "us/site" $out/wide_tl.log; python tools/trace_overlap.py $out/wide_tl/wide_kernel_trace.csv; echo; python tools/trace_timeline.py $out/wide_tl/wide_kernel_trace.csv | tail -24; } > $out/overlap.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wide_alone -o wide -- python tools/wide_bench.py 1000000 4096 none > $out/wide_alone.log 2>&1
# -matchDynamic: what runs beside what in the steady loop (10 batches before the last sweep launch)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/qs_tl -o qs -- python tools/qsweep_bench.py 1000000 10000 8192 > $out/qs_tl.log 2>&1 < /dev/null
{ grep matchDynamic $out/qs_tl.log; python tools/trace_busy.py $(find $out/qs_tl -name "*kernel_trace.csv" | head -1) @qss_sweep 10; } > $out/matchdynamic_busy.txt 2>&1
# what the consumers cost at the north-star width: the measurement build (wrong results) with the fill, its stores, the sweep's walks switched off
if [ -f pbwt_amd/libpbwtgpu_measure.so ]; then
  { for env in "X=1" "PBWTAMD_NOFILL=1" "PBWTAMD_DEBUG_FILL_NOWRITE=1" "PBWTAMD_DEBUG_SWEEP=2"; do echo "$env"; env PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_measure.so $env timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1; done
    echo "chain alone"; timeout 200 python tools/wide_bench.py 1000000 8192 none 2>&1 | tail -1; } > $out/consumer_pricing.txt 2>&1
fi
rm -f $out/*/*_kernel_trace.csv $out/*/*/*_kernel_trace.csv
find $out -name "*.csv" | head -40
du -sh $out
