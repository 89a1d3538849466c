# tools/many_panels_sweep.sh: bench.py --panels P over PANELS (default "1 2 4 6 8 10") on the bench option set; CHAIN_ONLY=1: without the consumers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
extra=""; [ -n "$CHAIN_ONLY" ] && extra="--no-within --no-pack3"
run() { timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu --no-1m 2>gpurun_out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('$*', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'us/launch %.2f' % d['roofline']['us_per_launch'])" || tail -5 gpurun_out/err.log; }
for P in ${PANELS:-1 2 4 6 8 10}; do run --panels $P $extra; done
