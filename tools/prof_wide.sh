mkdir -p gpurun_out/p1m; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_integration.py "tests/test_gpu_configs.py::test_wider_than_2_20_haplotypes" -x -q -m gpu 2>&1 | tail -6
timeout 900 python tools/qsweep_bench.py 1000000 10000 2048
timeout 900 python tools/qsweep_bench.py 100000 10000 4096
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/p1m/qsq -o w -- python tools/qsweep_bench.py 1000000 10000 1024 > gpurun_out/p1m/qsq.log 2>&1
python - <<'PY'
import csv, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/p1m/qsq/w_counter_collection.csv')):
    d[(r['Kernel_Name'][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted({k for k,_ in d}):
    if not any(x in k for x in ('qss','qs_')): continue
    w=sum(d[(k,'SQ_WAVES')])/len(d[(k,'SQ_WAVES')])
    g=lambda c: sum(d[(k,c)])/len(d[(k,c)])/w if d.get((k,c)) else 0
    print("%-46s waves %8.0f  per wave: VALU %7.0f SALU %7.0f VMEM_RD %6.0f cycles %9.0f wait %9.0f active %7.0f" % (k,w,g('SQ_INSTS_VALU'),g('SQ_INSTS_SALU'),g('SQ_INSTS_VMEM_RD'),g('SQ_WAVE_CYCLES'),g('SQ_WAIT_ANY'),g('SQ_ACTIVE_INST_ANY')))
PY
