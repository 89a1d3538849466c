import os, sys, time
os.environ["PBWTAMD_BUILD_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M, N = 100000, 16384
eng = amd.Engine(M, batch_sites=512)
buf = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=7, kind=0); eng.sync()
bits = buf.cpu().numpy().view(np.uint32)
want = os.environ.get("WANT_YZ", "1") == "1"
for i in range(3):
    t0 = time.perf_counter(); b = eng.build(bits, with_d=False, want_yz=want); dt = time.perf_counter() - t0
    print("call %d: %.2f ms" % (i, 1e3 * dt))
