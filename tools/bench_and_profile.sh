#!/bin/bash
# tools/bench_and_profile.sh <tag> — VERDICT r3 item 4: the bench line and the profiles it is judged against from ONE box in ONE gpurun.
#   1. python bench.py --steps 20 --warmup 5                      (the driver's command; the line goes to gpurun_out/<tag>/bench.json)
#   2. rocprofv3 --kernel-trace --stats around the same command without the secondary objects (--no-cpu --no-1m): per-kernel averages
#   3. the two PMC passes (FETCH_SIZE, WRITE_SIZE) + the calibration kernel, on a shorter run of the same path
#   4. profiles/<tag>_bench_and_profile.json: host, GPU serial, the line, the chain kernels' average durations, their sum per round of 8 sites,
#      roofline.frac recomputed from the CSV, and whether sum x rounds fits inside ms_per_step
tag=${1:-r04}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ hostname; cat /etc/machine-id 2>/dev/null; rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "serial\|unique" | head -4; } > $out/box.txt 2>&1
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
B="python bench.py --steps 20 --warmup 5 --no-cpu --no-1m"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $B > $out/trace.log 2>&1
S="python bench.py --steps 2 --warmup 1 --no-cpu --no-1m"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- $S > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- $S > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o bench -- $S > $out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calib_fetch -o calib -- ./tools/pmc_calib > $out/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/calib_write -o calib -- ./tools/pmc_calib > $out/calib_write.log 2>&1
# the north-star width, same box: kernel stats, traffic, SQ counters
WIDE="python tools/wide_bench.py 1000000 2048 hp"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wide_trace -o wide -- $WIDE > $out/wide_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/wide_fetch -o wide -- $WIDE > $out/wide_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/wide_write -o wide -- $WIDE > $out/wide_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/wide_sq -o wide -- $WIDE > $out/wide_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wide_alone -o wide -- python tools/wide_bench.py 1000000 4096 none > $out/wide_alone.log 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
python tools/reconcile_profile.py $tag
du -sh $out
