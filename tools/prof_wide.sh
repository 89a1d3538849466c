mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p1m/*
run() { echo "== $*"; env "$@" timeout 300 python tools/wide_bench.py 100000 16384 hp; }
run PBWTAMD_X=0
run PBWTAMD_S2_CUS=0
run PBWTAMD_S2_CUS=0 PBWTAMD_RANK_R4=2
run PBWTAMD_RANK_R4=2
run PBWTAMD_S2_CUS=192
run PBWTAMD_S2_CUS=128
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p1m/tr -o wide -- python tools/wide_bench.py 100000 8192 hp > gpurun_out/p1m/tr.log 2>&1; grep "us/site" gpurun_out/p1m/tr.log
python tools/trace_overlap.py gpurun_out/p1m/tr/wide_kernel_trace.csv
python tools/trace_timeline.py gpurun_out/p1m/tr/wide_kernel_trace.csv | tail -30
rm -rf gpurun_out/p1m/tr
