cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>gpurun_out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 gpurun_out/err.log; }
for P in 1 2 4 8; do run --panels $P --no-within --no-pack3; done
for P in 1 4 16 64; do run --haps 10000 --panels $P --no-within --no-pack3; done
for P in 1 4 16; do run --haps 25000 --panels $P; done
