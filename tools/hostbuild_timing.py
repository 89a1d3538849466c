"""phases of the host-buffer build entry point (PBWTAMD_BUILD_TIMING=1): python tools/hostbuild_timing.py [M] [sites]"""
import os, sys, time
os.environ["PBWTAMD_BUILD_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
eng = amd.Engine(M, batch_sites=512)
buf = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=7, kind=0); eng.sync()
bits = buf.cpu().numpy().view(np.uint32)
eng.build(bits[:1024], with_d=False)
for i in range(3):
    t0 = time.perf_counter(); b = eng.build(bits, with_d=False); dt = time.perf_counter() - t0
    print("call %d: %.2f ms = %.3e site*haps/s" % (i, 1e3 * dt, M * N / dt))
