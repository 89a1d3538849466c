mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
echo "R4=1"; timeout 300 python tools/wide_bench.py 1000000 4096 none; timeout 300 python tools/wide_bench.py 1000000 4096 hp
echo "R4=0"; PBWTAMD_RANK_R4=0 timeout 300 python tools/wide_bench.py 1000000 4096 none; PBWTAMD_RANK_R4=0 timeout 300 python tools/wide_bench.py 1000000 4096 hp
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p1m/t2 -o w -- python tools/wide_bench.py 1000000 2048 none > gpurun_out/p1m/t2.log 2>&1; grep "skel_" gpurun_out/p1m/t2/w_kernel_stats.csv | cut -c1-150
