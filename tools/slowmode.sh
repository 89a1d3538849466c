#!/bin/bash
# tools/slowmode.sh <reps>: the 100 k run repeated in fresh processes pinned to the CPUs of NUMA node 0, node 1, or unpinned (the slow level of DESIGN.md section 8)
# PINS entries: a taskset CPU list, "none" (the library places itself: pbwtamd_engine_create, round 5), "nopin" (PBWTAMD_PIN=0: nobody does)
out=gpurun_out/slowmode; mkdir -p $out
echo "GPU numa nodes: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"; lscpu | grep -i "numa node" 
for r in $(seq ${1:-6}); do
  for pin in ${PINS:-"0-63" "64-127" "none"}; do
    if [ $pin = none ]; then t=$(timeout 300 python tools/wide_bench.py 100000 131072 hp 2>&1 | tail -1 | sed -n 's/.*: \([0-9.]*\) us\/site.*chain \([0-9.]*\) us.*/\1 \2/p')
    elif [ $pin = nopin ]; then t=$(PBWTAMD_PIN=0 timeout 300 python tools/wide_bench.py 100000 131072 hp 2>&1 | tail -1 | sed -n 's/.*: \([0-9.]*\) us\/site.*chain \([0-9.]*\) us.*/\1 \2/p')
    else t=$(timeout 300 taskset -c $pin python tools/wide_bench.py 100000 131072 hp 2>&1 | tail -1 | sed -n 's/.*: \([0-9.]*\) us\/site.*chain \([0-9.]*\) us.*/\1 \2/p'); fi
    echo "pin $pin: $t"
  done
done | tee $out/runs.txt
python - <<'PY'
import collections
d = collections.defaultdict(list)
for l in open("gpurun_out/slowmode/runs.txt"):
    if l.startswith("pin"):
        p = l.split(); d[p[1]].append(float(p[2]))
for k, v in d.items(): print(k, "min %.3f max %.3f" % (min(v), max(v)), sorted(v))
PY
