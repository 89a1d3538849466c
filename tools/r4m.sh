#!/bin/bash
# baseline of the restored tree: full GPU suite + the two widths
out=gpurun_out/r4m; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{ echo "1M: $(timeout 200 python tools/wide_bench.py 1000000 8192 hp 2>&1 | tail -1)"
  echo "1M chain only: $(timeout 200 python tools/wide_bench.py 1000000 8192 none 2>&1 | tail -1)"
  echo "100k: $(timeout 200 python tools/wide_bench.py 100000 16384 hp 2>&1 | tail -1)"
  echo "100k iid: $(KIND=1 timeout 200 python tools/wide_bench.py 100000 8192 hp 2>&1 | tail -1)"
} > $out/ab.txt 2>&1
cat $out/ab.txt
