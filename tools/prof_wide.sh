mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p1m/full_a.log 2>&1; tail -3 gpurun_out/p1m/full_a.log | cut -c1-200
PBWTAMD_POISON=165 timeout 1500 python -m pytest tests -q --tb=line -m gpu > gpurun_out/p1m/full_poison.log 2>&1; tail -8 gpurun_out/p1m/full_poison.log | cut -c1-200
