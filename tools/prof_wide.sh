mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo base; timeout 300 python tools/wide_bench.py 100000 16384 hp
echo "chain on 160-255"; PBWTAMD_CHAIN_CU_LO=160 timeout 300 python tools/wide_bench.py 100000 16384 hp
echo "chain on 128-255, consumers 0-127"; PBWTAMD_CHAIN_CU_LO=128 PBWTAMD_S2_CUS=128 timeout 300 python tools/wide_bench.py 100000 16384 hp
echo "chain on 192-255, consumers 0-191"; PBWTAMD_CHAIN_CU_LO=192 PBWTAMD_S2_CUS=192 timeout 300 python tools/wide_bench.py 100000 16384 hp
done
echo "chain-only masks"; timeout 300 python tools/wide_bench.py 100000 16384 none; PBWTAMD_CHAIN_CU_LO=160 timeout 300 python tools/wide_bench.py 100000 16384 none; PBWTAMD_CHAIN_CU_LO=192 timeout 300 python tools/wide_bench.py 100000 16384 none
