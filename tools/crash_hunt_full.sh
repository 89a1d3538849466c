#!/bin/bash
# tools/crash_hunt_full.sh <tag> <iterations>: the whole -m gpu suite under rocgdb until a run dies; backtraces to gpurun_out/<tag>/
tag=${1:-r5z}; n=${2:-2}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq $n); do
  timeout 2400 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGSEGV nostop noprint pass" -ex run -ex "bt 40" -ex "thread apply all bt 25" --args python -m pytest tests -x -q -m gpu > $out/run$i.log 2>&1
  if grep -q " passed" $out/run$i.log && ! grep -q "SIGABRT\|Aborted\|failed" $out/run$i.log; then echo "run $i: $(grep ' passed' $out/run$i.log | tail -1)"; else echo "run $i DIED"; grep -n "SIGABRT\|Aborted\|signal\|^#[0-9]\|Thread [0-9]" $out/run$i.log | head -120 | cut -c1-240; break; fi
done
