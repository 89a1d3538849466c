#!/bin/bash
# tools/c5_full.sh <tag> [sites=10000000]: configs[4] at its own length in ONE streamed pass (build + maxWithin + pack3 + -matchDynamic, 10 000 queries)
tag=${1:-r5n}; sites=${2:-10000000}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py --stream-panel --with-queries --ns-sites $sites > $out/c5_full.json 2> $out/c5_full.err; tail -2 $out/c5_full.err; cut -c1-1200 $out/c5_full.json
timeout 300 python bench.py --stream-panel --with-queries --ns-sites 65536 > $out/c5_first65536.json 2>> $out/c5_full.err; cut -c1-700 $out/c5_first65536.json
