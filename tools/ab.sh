#!/bin/bash
# tools/ab.sh <out> <M> <sites> <reps> "<label>=<ENV...>" ... : interleaved repeats of tools/wide_bench.py, the minimum and the median per variant
out=$1; M=$2; sites=$3; reps=$4; shift 4
mkdir -p $(dirname $out)
for r in $(seq $reps); do
  for v in "$@"; do
    lab=${v%%=*}; envs=${v#*=}
    t=$(env $envs timeout 300 python tools/wide_bench.py $M $sites hp 2>&1 | tail -1 | sed -n 's/.*: \([0-9.]*\) us\/site.*/\1/p')
    echo "$lab $t"
  done
done > $out.raw
python - $out.raw $M <<'PY' | tee -a $out
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 2: d[p[0]].append(float(p[1]))
for k, v in d.items():
    v.sort(); print("M %s %-14s min %.3f median %.3f  (%s)" % (sys.argv[2], k, v[0], v[len(v) // 2], " ".join("%.3f" % x for x in v)))
PY
