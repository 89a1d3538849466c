"""-matchDynamic path: Mq query haplotypes against a packed panel of M; queries and panel are the two parts of ONE
synthetic founder-mosaic panel (shared founders: matches run for many sites, as with real data), us per site"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pbwt_amd as amd
M, Mq, N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000, int(sys.argv[3]) if len(sys.argv) > 3 else 4096
full = amd.Engine(M + Mq, batch_sites=512)
buf = torch.zeros((N, full.wpc), dtype=torch.int32, device="cuda")
full.synth_device(buf.data_ptr(), 0, N, seed=3, kind=0); full.sync()
bits = buf.cpu().numpy().view(np.uint32)
hap = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :M + Mq]
def pack(h):
    m = h.shape[1]; wpc = amd.wpc_for(m)
    out = np.zeros((N, wpc * 4), np.uint8)
    pb = np.packbits(h, axis=1, bitorder="little"); out[:, :pb.shape[1]] = pb
    return out.view(np.uint32)
ep = amd.Engine(M, batch_sites=512); eq = amd.Engine(Mq, batch_sites=512)
pz = ep.build(pack(hap[:, :M]), with_d=False)["yz"]; qz = eq.build(pack(hap[:, M:]), with_d=False)["yz"]
for rep in range(2):
    t0 = time.perf_counter(); tc0 = time.process_time()
    recs, nom, tot = ep.match_sweep(pz, N, qz, Mq)
    dt = time.perf_counter() - t0; print("host cpu time %.3f s of %.3f wall" % (time.process_time() - tc0, dt))
    print("matchDynamic %d x %d queries x %d sites: %.1f ms = %.2f us/site, %d records, %.3e panel site*haps/s" % (M, Mq, N, 1e3 * dt, 1e6 * dt / N, len(recs), M * N / dt))
np.savez("/tmp/qsweep_case.npz", pz=pz, qz=qz) if os.path.isdir("/tmp") and os.environ.get("QS_SAVE") else None
