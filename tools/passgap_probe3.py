import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M, N = 10000, 100000
eng = amd.Engine(M, batch_sites=512)
buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
eng.synth_device(buf.data_ptr(), 0, N, seed=0x5EED + M, kind=0); eng.sync()
bits = buf.cpu().numpy().view(np.uint32)
yz = eng.build(bits, with_d=False)["yz"]
eng.max_within(yz, N, mode="hist")
for i in range(2):
    t0 = time.perf_counter(); rec = eng.max_within(yz, N, mode="records"); t1 = time.perf_counter()
    eng.pass_begin(8); t2 = time.perf_counter()
    print("records %d in %.3f s; the next pass_begin %.1f ms" % (len(rec), t1 - t0, 1e3 * (t2 - t1)))
    del rec
