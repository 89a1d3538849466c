mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_cli.py tests/test_integration.py -x -q -m gpu 2>&1 | tail -4
PBWTAMD_TRACE_QS=1 timeout 900 python tools/qsweep_bench.py 1000000 10000 8192 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p1m/t2 -o w -- python tools/wide_bench.py 1000000 2048 hp > gpurun_out/p1m/t2.log 2>&1; grep "pack3\|fill\|sweep" gpurun_out/p1m/t2/w_kernel_stats.csv | cut -c1-150
python bench.py --steps 20 --warmup 2 --own-stream --no-cpu --no-1m | python -c "import json,sys; d=json.load(sys.stdin); print('own-stream', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 2 --no-cpu --no-1m | python -c "import json,sys; d=json.load(sys.stdin); print('torch-stream', d['value'], d['ms_per_step'])"
