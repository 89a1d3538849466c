import os, sys
os.environ["PBWTAMD_SKEL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd
M, N, B = int(sys.argv[1]), 2048, 512
eng = pbwt_amd.Engine(M, batch_sites=B)
buf = torch.zeros((N + 8, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(buf.data_ptr(), 0, N, seed=9, kind=0); eng.sync()
eng.pass_begin(N)
eng.pass_advance(buf.data_ptr(), N, N, pbwt_amd.OPT_WITH_D)
eng.pass_end(pbwt_amd.OPT_WITH_D)
ms, n = eng.chain_timing()
print(M, "%.2f us/site" % (1e3 * ms / N))
