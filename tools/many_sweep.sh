#!/bin/bash
# tools/many_sweep.sh <tag>: eight (six) panels of 100 k haplotypes through pbwtamd_pass_advance_many by chain form — three launches with grid.y = panel, the one-launch
# round with grid.y = panel, the team-persistent chain (panel p on XCD p) by team size — chain only and with the bench consumers
tag=${1:-r5k}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env $ENVS timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$ENVS $*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
{
for extra in "--no-within --no-pack3" ""; do
  for P in 6 8; do
    ENVS="PBWTAMD_ONEPASS=0 PBWTAMD_TEAM=0"; run --panels $P $extra
    ENVS="PBWTAMD_ONEPASS=1 PBWTAMD_TEAM=0"; run --panels $P $extra
  done
  for K in ${KS8:-33 49 66 98}; do ENVS="PBWTAMD_ONEPASS=0 PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=$K"; run --panels 8 $extra; done
done
ENVS="PBWTAMD_ONEPASS=0 PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=66 PBWTAMD_TEAM_SYNC=0"; run --panels 8
} > $out/many.txt 2>&1
cat $out/many.txt
timeout 900 python -m pytest tests/test_gpu_z_fullsize.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
