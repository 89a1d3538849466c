#!/bin/bash
out=gpurun_out/r5i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PBWTAMD_TRACE_QS=1 timeout 150 python tools/md_bench.py 16384 1 2>&1 | grep -v amdgpu.ids | tail -4
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o md -- python tools/md_bench.py 16384 1 > $out/tr.log 2>&1
python tools/md_gap_detail.py $(find $out/tr -name "*kernel_trace.csv" | head -1) 30 | tee $out/detail.txt | head -70
rm -rf $out/tr
