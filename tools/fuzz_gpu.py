"""open-ended randomised differential run of the device paths against the oracle: python tools/fuzz_gpu.py SEED SECONDS
(the fixed-budget version runs in the suite: tests/test_gpu_fuzz.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import run_cases
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time()
n, bad = run_cases(seed, budget_s=budget, verbose=True)
print("fuzz seed %d: %d cases, %d bad, %.0fs" % (seed, n, len(bad), time.time() - t0))
