# width sweep of the bench path on the GPU box: us/site end to end (build + maxWithin histogram + pack3) and chain alone
# usage (from the repo root): gpurun -- 'bash tools/prof_wide.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 10000 40000 100000 150000 250000 400000 600000 1000000; do
  timeout 300 python tools/wide_bench.py $M 16384 hp
  timeout 300 python tools/wide_bench.py $M 16384 none
done
