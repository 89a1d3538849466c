import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbwt_amd as amd, oracle as orc
for M, N, B in [(1, 16, 8), (2, 16, 8), (3, 24, 8), (64, 16, 8), (65, 16, 8)]:
    eng = amd.Engine(M, batch_sites=B)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    eng.synth_device(buf.data_ptr(), 0, N, seed=1000 + M, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM
    eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
    ca, cd, _ = eng.get_checksums(0, N + 1)
    print(M, N, B, "a bad sites", np.nonzero(ca != o["csum_a"])[0].tolist(), "d bad sites", np.nonzero(cd != o["csum_d"])[0].tolist())
    a, d = eng.get_state()
    print("   final", np.array_equal(a, o["aFend"]), np.array_equal(d, o["d_final"]), d[:4], o["d_final"][:4])
