"""engine create/destroy churn (streams with CU masks, events, pinned buffers) beside torch copies — looks for the rare abort seen in the GPU suite"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import pbwt_amd.api as amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(n):
    M = (70001, 300000, 3000, 1025)[i % 4]
    N, batch = 40, 16
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=1000 + M + i, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST
    eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
    a, d = eng.get_state()
    del eng, buf
    if i % 50 == 0: print("iter", i, flush=True)
print("churn ok", n)
