"""phase timing of the step kernel (PBWTAMD_PROFILE=1): where does a launch spend its time?"""
import os, sys
os.environ["PBWTAMD_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = 2048
eng = pbwt_amd.Engine(M, batch_sites=B)
buf = torch.zeros((N + 1, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N + 1, seed=3, kind=0); eng.sync()
for opts in (pbwt_amd.OPT_WITH_D, 0):
    eng.pass_begin(N + 1)
    eng.pass_advance(buf.data_ptr(), N, N + 1, opts); eng.sync()
    ms, n = eng.chain_timing(); ns = eng.chain_sites()
    pr = eng.phase_profile()
    d = (pr[:, 1:7] - pr[:, 0:1]) * 10.0     # ns since kernel entry of that tile (100 MHz clock)
    print("M=%d B=%d with_d=%d: %.2f us/launch over %d launches, %.2f us/site" % (M, B, bool(opts), 1e3 * ms / n, n, 1e3 * ms / ns))
    print("  phase stamps (ns after tile entry), median over tiles: " + "  ".join("%d:%.0f" % (i + 1, np.median(d[:, i])) for i in range(6)))
    print("  tile entry spread: %.0f ns, last tile exit - first tile entry: %.0f ns" % ((pr[:, 0].max() - pr[:, 0].min()) * 10.0, (pr[:, 6].max() - pr[:, 0].min()) * 10.0))
