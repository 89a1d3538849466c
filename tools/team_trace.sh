#!/bin/bash
# tools/team_trace.sh <tag>: kernel timeline of the team-persistent chain with eight panels (what runs between two batches' launches, and how long the launches take)
tag=${1:-r5c}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --panels 8 --haps 65536 --steps 1 --warmup 1 --no-cpu --no-1m"
PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t8 -o t -- $B --no-within --no-pack3 > $out/t8.log 2>&1
python tools/trace_dump.py $(find $out/t8 -name "*kernel_trace.csv" | head -1) skel_team 80 > $out/t8_chain_only.txt 2>&1
PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=96 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t8c -o t -- $B > $out/t8c.log 2>&1
python tools/trace_dump.py $(find $out/t8c -name "*kernel_trace.csv" | head -1) skel_team 200 > $out/t8_consumers.txt 2>&1
PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t1c -o t -- python tools/wide_bench.py 65536 2048 hp > $out/t1c.log 2>&1
python tools/trace_dump.py $(find $out/t1c -name "*kernel_trace.csv" | head -1) skel_team 100 > $out/t1_consumers_K128.txt 2>&1
tail -2 $out/t8.log $out/t8c.log $out/t1c.log
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
head -70 $out/t8_chain_only.txt
