#!/bin/bash
# tools/stream_q.sh <tag> [sites]: the streamed query sweep — parity tests, then configs[4] whole in one streamed pass over <sites> sites (default 65 536), and the
# resident / packed-panel figures of today's match_dynamic for comparison
tag=${1:-r5m}; sites=${2:-65536}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "match_sweep" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 1200 python bench.py --stream-panel --with-queries --ns-sites $sites > $out/c5q.json 2> $out/c5q.err; tail -2 $out/c5q.err; cut -c1-900 $out/c5q.json
timeout 1200 python bench.py --stream-panel --with-queries --ns-no-pack3 --ns-sites $sites > $out/c5q_nopack3.json 2>> $out/c5q.err; cut -c1-400 $out/c5q_nopack3.json
