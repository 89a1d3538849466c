#!/bin/bash
# tools/profile_round.sh <tag> — rocprofv3 evidence for one round, written under gpurun_out/<tag>/
# (kernel-trace stats and the two PMC passes are separate runs, as the pool requires)
tag=${1:-r01}
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
find $out -name "*.csv" | head -20
du -sh $out
