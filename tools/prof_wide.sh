mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p1m/full_d$i.log 2>&1; tail -2 gpurun_out/p1m/full_d$i.log | cut -c1-200
done
