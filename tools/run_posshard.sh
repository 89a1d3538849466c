#!/bin/bash
# tools/run_posshard.sh WORLD M N [B] [KIND] [STEP] [CSUM] — the position-sharded worker with WORLD ranks on this box's GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=$1; export PS_M=$2 PS_N=$3 PS_B=${4:-512} PS_KIND=${5:-0} PS_STEP=${6:-8192} PS_CSUM=${7:-1}
export OUT_DIR=gpurun_out/r3/ps_${W}_${PS_M}_${PS_N}; mkdir -p $OUT_DIR
timeout ${PS_TIMEOUT:-300} python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((20000 + RANDOM % 20000)) tests/posshard_worker.py > $OUT_DIR/log.txt 2>&1
echo "rc=$? world=$W M=$PS_M N=$PS_N"; tail -5 $OUT_DIR/log.txt | cut -c1-400; cat $OUT_DIR/ps*.json 2>/dev/null; echo
