mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== k2 wide"; timeout 300 python tools/wide_bench.py 1000000 2048 none; echo rc=$?
PBWTAMD_K2_WIDE=0 timeout 300 python tools/wide_bench.py 1000000 2048 none
timeout 300 python tools/wide_bench.py 600000 2048 none; PBWTAMD_K2_WIDE=0 timeout 300 python tools/wide_bench.py 600000 2048 none
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/wide_bench.py 1000000 2048 hp
timeout 300 python tools/wide_bench.py 100000 16384 hp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p1m/t2 -o w -- python tools/wide_bench.py 1000000 2048 hp > gpurun_out/p1m/t2.log 2>&1; grep "sweep_\|skel_\|pack3" gpurun_out/p1m/t2/w_kernel_stats.csv | cut -c1-150
