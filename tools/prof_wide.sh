mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/p1m/bench_qs.json 2> gpurun_out/p1m/bench_qs.err; tail -3 gpurun_out/p1m/bench_qs.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/p1m/bench_qs.json').read().strip().splitlines()[-1])
print(json.dumps(d.get("match_dynamic"), indent=1))
PY
