"""is the chain paced by the host's launch rate?  time the enqueue (pass_advance returns) vs the GPU completion"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd
M, S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 8192, int(os.environ.get('HB_B', '512'))
eng = pbwt_amd.Engine(M, batch_sites=B)
n_total = 6 * S
panel = torch.empty((n_total, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(panel.data_ptr(), 0, n_total, seed=5, kind=int(os.environ.get('HB_KIND', '0'))); eng.sync()
opts = pbwt_amd.OPT_WITH_D | (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng.pass_begin(n_total)
rb = eng.wpc * 4
for i in range(6):
    k = i * S
    t0 = time.perf_counter()
    eng.pass_advance(panel.data_ptr() + k * rb, S, min(S + 8, n_total - k), opts)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print("step %d: enqueue %.2f ms (%.2f us/site), complete %.2f ms (%.2f us/site)" % (i, 1e3 * (t1 - t0), 1e6 * (t1 - t0) / S, 1e3 * (t2 - t0), 1e6 * (t2 - t0) / S))
