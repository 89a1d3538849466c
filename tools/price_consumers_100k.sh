cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu --no-1m $1 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('%-28s' % ('$1'), 'us/site %.3f' % (1e3*d['ms_per_step']/8192), 'us/launch %.2f' % d['roofline']['us_per_launch'])"; }
for i in 1 2 3; do run ""; run "--no-pack3"; run "--no-within"; run "--no-within --no-pack3"; done
