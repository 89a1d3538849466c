"""host entry point pbwtamd_build (columns in host memory -> .pbwt bytes): A-only (the reference's -readMacs path) vs AD"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 16384
eng = amd.Engine(M, batch_sites=512)
buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=3, kind=0); eng.sync()
bits = buf.cpu().numpy().view(np.uint32)
for wd in (False, True, False, True):
    t0 = time.perf_counter()
    b = eng.build(bits, with_d=wd)
    dt = time.perf_counter() - t0
    print("build with_d=%d: %.1f ms = %.2f us/site, %.3e site*haps/s" % (wd, 1e3 * dt, 1e6 * dt / N, M * N / dt))
