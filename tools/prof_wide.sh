mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p1m/*
for J in 1 0; do
PBWTAMD_QS_JUMP8=$J timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p1m/q$J -o qs -- python tools/qsweep_bench.py 1000000 10000 4096 > gpurun_out/p1m/q$J.log 2>&1
echo "JUMP8=$J"; grep "matchDynamic" gpurun_out/p1m/q$J.log; head -8 gpurun_out/p1m/q$J/qs_kernel_stats.csv | cut -c1-120
rm -f gpurun_out/p1m/q$J/qs_kernel_trace.csv
done
