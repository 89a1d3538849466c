#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1000000 100000; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl_$M -o w -- python tools/wide_bench.py $M 8192 hp > $out/tl_$M.log 2>&1
tail -1 $out/tl_$M.log
f=$(find $out/tl_$M -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f | tee $out/gaps_$M.txt
rm -rf $out/tl_$M
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl_alone -o w -- python tools/wide_bench.py 1000000 8192 none > $out/tl_alone.log 2>&1
f=$(find $out/tl_alone -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f | tee $out/gaps_alone.txt
rm -rf $out/tl_alone
