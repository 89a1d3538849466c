#!/bin/bash
# tools/quick_check.sh <tag> [pytest -k expression]: a subset of the -m gpu suite, then us/site at 100 k / 30 k with the bench consumers (three repeats)
tag=${1:-r5q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${2:-onepass or team or many_panels}" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{ for M in ${WIDTHS:-100000 30000 10000}; do for W in none hp; do for i in 1 2 3; do timeout 200 python tools/wide_bench.py $M 16384 $W 2>&1 | tail -1; done; done; done; } > $out/wb.txt 2>&1; cat $out/wb.txt
