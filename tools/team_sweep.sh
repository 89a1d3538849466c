#!/bin/bash
# tools/team_sweep.sh <tag>: the team-persistent chain (PBWTAMD_TEAM=1, skel_team_kernel) against the three-launch round — parity first, then us/site of ONE
# panel (chain alone and with the bench consumers) by team size K, then P = 8 panels through pbwtamd_pass_advance_many.  Output: gpurun_out/<tag>/
tag=${1:-r5a}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team or many_panels" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
fi
wb() { env "$@" timeout 200 python tools/wide_bench.py $M $S $W 2>&1 | tail -1; }
{
for M in ${WIDTHS:-100000}; do
  S=16384
  for W in none hp; do
    echo "== M $M $W three launches"; wb PBWTAMD_TEAM=0; wb PBWTAMD_TEAM=0
    for K in ${KS:-49 66 98 128}; do echo "== M $M $W team K=$K"; wb PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=$K; done
  done
done
} > $out/one_panel.txt 2>&1
cat $out/one_panel.txt
run() { env $ENVS timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$ENVS $*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
{
for extra in "--no-within --no-pack3" ""; do
  ENVS="PBWTAMD_TEAM=0"; run --panels 8 $extra; run --panels 6 $extra
  for K in ${KS8:-49 66 98}; do ENVS="PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=$K"; run --panels 8 $extra; done
done
} > $out/many_panels.txt 2>&1
cat $out/many_panels.txt
