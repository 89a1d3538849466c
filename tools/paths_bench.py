"""absolute times of the host entry points at M = 100k (packed panel in host memory), to compare with the reference CPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = amd.Engine(M, batch_sites=512)
buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=3, kind=0); eng.sync()
bits = buf.cpu().numpy().view(np.uint32)
yz = eng.build(bits, with_d=False)["yz"]
np.save("gpurun_out/paths_yz.npy", yz) if os.environ.get("SAVE_YZ") else None
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print("%-28s %8.1f ms  %s" % (name, 1e3 * dt, r))
t("maxWithin hist", lambda: int(eng.max_within(yz, N, mode="hist").sum()))
t("maxWithin records", lambda: len(eng.max_within(yz, N, mode="records")))
t("longWithin L=200", lambda: len(eng.long_within(yz, N, 200)))
t("longWithin L=1000", lambda: len(eng.long_within(yz, N, 1000)))
t("haplotypes", lambda: eng.haplotypes(yz, N).shape)
t("sweep_AD checksums", lambda: len(eng.sweep_AD(yz, N)["csum_a"]))
