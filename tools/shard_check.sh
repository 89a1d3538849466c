#!/bin/bash
# tools/shard_check.sh <tag>: the position-sharded chain — parity with 2-8 ranks on this box's one GPU, then ONE rank in shard mode at 1 M haplotypes (the price of the
# structure) with the local and the two-pass tile scan
tag=${1:-r5s}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "position_sharded_chain" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{ for L in 1 0; do echo "== PBWTAMD_K2S_LOCAL=$L"; PBWTAMD_K2S_LOCAL=$L bash tools/run_posshard_bench.sh 1 1000000 4; done; echo "== tile tables recomputed by the consumers (PBWTAMD_SHARD_TABLES=0)"; PBWTAMD_SHARD_TABLES=0 bash tools/run_posshard_bench.sh 1 1000000 4
  echo "== plain engine, same width"; timeout 200 python tools/wide_bench.py 1000000 32768 hp 2>&1 | tail -1; } > $out/onerank.txt 2>&1; cat $out/onerank.txt
