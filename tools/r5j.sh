#!/bin/bash
out=gpurun_out/r5j; mkdir -p $out
timeout 2000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
KIND=1 bash tools/ab.sh $out/ab_100k_iid.txt 100000 16384 2 "p16=X=1" "p32=PBWTAMD_P16=0"
bash tools/ab.sh $out/ab_100k.txt 100000 131072 2 "p16=X=1" "p32=PBWTAMD_P16=0"
bash tools/ab.sh $out/ab_1m.txt 1000000 8192 2 "p16=X=1"
