"""what the first call after a pass costs (HIP runtime work deferred behind a pass): python tools/passgap_probe.py [M] [sites]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd as amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
eng = amd.Engine(M, batch_sites=512)
buf = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=7, kind=0); eng.sync()
opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
for i in range(4):
    t0 = time.perf_counter(); eng.pass_begin(N); t1 = time.perf_counter()
    eng.pass_advance(buf.data_ptr(), N, N, opts); t2 = time.perf_counter()
    eng.pass_end(opts); t3 = time.perf_counter()
    print("pass %d: pass_begin %.2f ms, advance (enqueue) %.2f, pass_end %.2f" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
