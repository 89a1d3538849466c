#!/bin/bash
# kernel durations of the fused many-panel chain at several panel counts (measurement aid; output under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
prof() { tag=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o mp -- python bench.py "$@" --no-within --no-pack3 --steps 2 --warmup 1 --no-cpu --no-1m > /dev/null 2>&1 < /dev/null
  f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag $*"; [ -n "$f" ] && head -7 "$f" | cut -c1-220
  rm -rf gpurun_out/$tag; }
prof a16 --haps 10000 --panels 16
prof a64 --haps 10000 --panels 64
prof b4 --panels 4
prof b8 --panels 8
