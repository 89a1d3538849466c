mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py "tests/test_gpu_configs.py::test_config4_shape_million_haplotypes" -x -q -m gpu 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p1m/t2 -o w -- python tools/wide_bench.py 1000000 2048 none > gpurun_out/p1m/t2.log 2>&1; grep "us/site" gpurun_out/p1m/t2.log; grep "skel_\|transpose" gpurun_out/p1m/t2/w_kernel_stats.csv | cut -c1-150
timeout 300 python tools/wide_bench.py 1000000 4096 none
timeout 300 python tools/wide_bench.py 1000000 4096 hp
timeout 300 python tools/wide_bench.py 100000 16384 none
timeout 300 python tools/wide_bench.py 100000 16384 hp
timeout 300 python tools/wide_bench.py 10000 65536 hp
