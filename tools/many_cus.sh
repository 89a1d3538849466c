#!/bin/bash
# tools/many_cus.sh <tag>: eight panels of 100 k beside the bench consumers by the number of CUs the consumer streams are confined to (measurement build, PBWTAMD_S2_CUS; 0 = no mask)
tag=${1:-r6c}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PBWTAMD_LIB=$PWD/pbwt_amd/libpbwtgpu_measure.so
{ for i in 1 2; do for C in ${CUS:-160 0 192 224 128}; do echo -n "S2_CUS=$C "; PBWTAMD_S2_CUS=$C timeout 300 python bench.py --panels 8 --steps 4 --warmup 1 --no-cpu --no-1m 2>$out/err.log | python -c "
import sys, json; d=json.loads(sys.stdin.readline()); print('value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"; done; done; } > $out/cus.txt 2>&1; cat $out/cus.txt
