#!/bin/bash
# tools/many_repeat.sh <tag>: eight panels of 100 k through pbwtamd_pass_advance_many, shipped defaults (team-persistent chain) against three launches with grid.y = panel, three repeats each
tag=${1:-r5t}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env $ENVS timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$ENVS $*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'])" || tail -5 $out/err.log; }
{ for i in 1 2 3; do
  ENVS="X=1"; run --panels 8
  ENVS="PBWTAMD_TEAM=0"; run --panels 8
  ENVS="PBWTAMD_TEAM=0"; run --panels 6
  ENVS="PBWTAMD_TEAM=1"; run --panels 6
  ENVS="PBWTAMD_TEAM_K=98"; run --panels 8
done; } > $out/many.txt 2>&1; cat $out/many.txt
