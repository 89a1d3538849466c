cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for G in 1 0; do echo "GRAPH=$G"; for M in 10000 100000 250000; do PBWTAMD_SKEL_GRAPH=$G python tools/hostrate.py $M 8192; for O in none hp; do PBWTAMD_SKEL_GRAPH=$G timeout 300 python tools/wide_bench.py $M 16384 $O; done; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
