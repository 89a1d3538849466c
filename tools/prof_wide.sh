mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq 1 4); do
timeout 900 python -m pytest tests -x -q -s -m gpu > gpurun_out/p1m/loop.log 2>&1 || { echo "FAILED at iteration $i"; grep -v "^  File" gpurun_out/p1m/loop.log | cut -c1-400 | tail -40; cp gpurun_out/p1m/loop.log gpurun_out/p1m/loop_fail.log; break; }
done
echo "done $i"; tail -2 gpurun_out/p1m/loop.log
