mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -2 gpurun_out/bench1.err; python -c "
import json; d=json.load(open('gpurun_out/bench1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch']); print(d.get('north_star_width')); print(d.get('cpu_baseline')); print(d.get('host_entry_points'))"
