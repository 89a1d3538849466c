#!/bin/bash
# tools/team_probe.sh <tag>: where a round of the team-persistent chain spends its time (PBWTAMD_TEAM_PROF stamps), one tile per member against two, nt against
# sc1 loads, and how the teams slow each other down when several XCDs are busy (P = 1, 2, 4, 8).  Output: gpurun_out/<tag>/
tag=${1:-r5b}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team or many_panels" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
fi
wb() { env "$@" timeout 200 python tools/wide_bench.py $M $S $W 2>&1 | tail -${TL:-1}; }
{
S=2048; W=none; TL=3
for cfg in "100000 128" "100000 98" "65536 128" "32768 128" "49152 96"; do set -- $cfg; M=$1
  echo "== prof M $M K $2"; wb PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=$2 PBWTAMD_TEAM_PROF=1
  echo "== prof sc1 M $M K $2"; wb PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_sc1.so PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=$2 PBWTAMD_TEAM_PROF=1
done
S=2048; W=hp; M=65536; echo "== prof hp M $M K 128"; wb PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128 PBWTAMD_TEAM_PROF=1
} > $out/prof.txt 2>&1
cat $out/prof.txt
{
S=16384; TL=1
for M in 65536 32768; do
  for W in none hp; do
    echo "== M $M $W three launches"; wb PBWTAMD_TEAM=0; wb PBWTAMD_TEAM=0
    echo "== M $M $W team K=128"; wb PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128; wb PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128
    echo "== M $M $W team sc1 K=128"; wb PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_sc1.so PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128
  done
done
} > $out/one_tile.txt 2>&1
cat $out/one_tile.txt
run() { env $ENVS timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-cpu --no-1m --no-within --no-pack3 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$ENVS $*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/round %.2f' % d['roofline']['us_per_launch'])" || tail -5 $out/err.log; }
{
for P in 1 2 4 8; do ENVS="PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=98"; run --panels $P; done
for P in 1 8; do ENVS="PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128"; run --panels $P --haps 65536; done
for P in 1 8; do ENVS="PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_sc1.so PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128"; run --panels $P --haps 65536; done
ENVS="PBWTAMD_TEAM=1 PBWTAMD_TEAM_K=128 PBWTAMD_TEAM_PROF=1"; run --panels 8 --haps 65536; grep "team prof" $out/err.log | tail -3
} > $out/pscale.txt 2>&1
cat $out/pscale.txt
