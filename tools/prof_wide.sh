mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p1m/full_b.log 2>&1; tail -3 gpurun_out/p1m/full_b.log | cut -c1-200
timeout 300 python tools/wide_bench.py 1000000 8192 hp
