import os, sys
os.environ["PBWTAMD_SKEL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd, oracle
for (M, N, B) in [(3000, 64, 32), (100000, 128, 64), (1500, 40, 8), (70000, 256, 256), (100000, 2048, 512), (1024, 64, 64), (1025, 72, 24), (10000, 512, 512), (33000, 512, 512), (1000000, 64, 32)]:
    eng = pbwt_amd.Engine(M, batch_sites=B)
    buf = torch.zeros((N + 8, eng.wpc), dtype=torch.int32, device="cuda")
    eng.synth_device(buf.data_ptr(), 0, N, seed=9, kind=0); eng.sync()
    bits = buf[:N].cpu().numpy().view(np.uint32)
    o = oracle.build_bitcols(bits, M, with_d=True, dump_sites=[N])
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, pbwt_amd.OPT_WITH_D)
    eng.pass_end(pbwt_amd.OPT_WITH_D)
    a, d = eng.get_state()
    ok = np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    ms, n = eng.chain_timing()
    print(M, N, B, "OK" if ok else "MISMATCH a:%d d:%d" % ((a != o["aFend"]).sum(), (d != o["d_final"]).sum()), "%.2f us/site" % (1e3 * ms / N), n)
