mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p1m/full_c.log 2>&1; tail -3 gpurun_out/p1m/full_c.log | cut -c1-200
PBWTAMD_POISON=77 timeout 600 python -m pytest tests/test_gpu_z_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "config4 or north_star or 300000 or wider" 2>&1 | tail -2
