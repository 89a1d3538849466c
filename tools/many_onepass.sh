#!/bin/bash
# tools/many_onepass.sh <tag>: eight panels of 100 k through pbwtamd_pass_advance_many — the one-launch round with grid.y = panel for ALL eight (PBWTAMD_MANY_ONEPASS_MAX=4096)
# against the team-persistent chain (default) and the three-launch round; chain only and with the bench consumers
tag=${1:-r5mo}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env $ENVS timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$ENVS $*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
{
for extra in "--no-within --no-pack3" ""; do for P in ${PS:-8 6}; do for i in 1 2; do
    ENVS="PBWTAMD_TEAM=0 PBWTAMD_MANY_ONEPASS_MAX=4096"; run --panels $P $extra
    ENVS="PBWTAMD_TEAM=-1"; run --panels $P $extra
    ENVS="PBWTAMD_TEAM=0 PBWTAMD_ONEPASS=0"; run --panels $P $extra
done; done; done
} > $out/many.txt 2>&1
cat $out/many.txt
if [ -n "$TESTS" ]; then PBWTAMD_MANY_ONEPASS_MAX=4096 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_panels" 2>&1 | tail -2; fi
