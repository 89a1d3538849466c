mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "600100" --durations=5 2>&1 | tail -12
