#!/bin/bash
# tools/crash_hunt.sh <tag> <iterations> <pytest -k expr>: a subset of the -m gpu suite repeated under rocgdb until a run dies; the backtrace of every thread goes to gpurun_out/<tag>/
tag=${1:-r5y}; n=${2:-6}; expr=${3:-"both_chains_every_site"}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq $n); do
  timeout 900 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGSEGV nostop noprint pass" -ex run -ex "bt 40" -ex "thread apply all bt 25" --args python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$expr" > $out/run$i.log 2>&1
  if grep -q "passed" $out/run$i.log && ! grep -q "SIGABRT\|Aborted\|failed" $out/run$i.log; then echo "run $i: $(grep passed $out/run$i.log | tail -1)"; rm -f $out/run$i.log; else echo "run $i DIED"; grep -n "SIGABRT\|Aborted\|signal\|#[0-9]" $out/run$i.log | head -80 | cut -c1-220; break; fi
done
