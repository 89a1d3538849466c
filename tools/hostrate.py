"""host enqueue time of the chain against its device time: python tools/hostrate.py M sites (PBWTAMD_THR_ROUNDS=0 to lift the throttle)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd as amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
B = 512
eng = amd.Engine(M, batch_sites=B)
N = sites + B
panel = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(panel.data_ptr(), 0, N, seed=1, kind=0); eng.sync()
opts = amd.OPT_WITH_D
eng.pass_begin(N)
eng.pass_advance(panel.data_ptr(), B, B + 8, opts); eng.sync()
t0 = time.perf_counter()
eng.pass_advance(panel.data_ptr() + B * eng.wpc * 4, sites, sites, opts)
t1 = time.perf_counter()
eng.pass_end(opts)
t2 = time.perf_counter()
nl = 3 * sites // 8
print("M %d: host enqueue %.2f us/launch (returned after %.1f ms), device done after %.1f ms = %.2f us/launch" % (M, 1e6 * (t1 - t0) / nl, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e6 * (t2 - t0) / nl))
