"""one pass of the bench path at a given width (default: the north-star width, 1 M haplotypes) for rocprofv3:
python tools/wide_bench.py [M] [sites] [opts: h=hist p=pack3 c=checksum none=chain only]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd as amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
what = sys.argv[3] if len(sys.argv) > 3 else "hp"
B = int(os.environ.get("WB_BATCH", "512"))
eng = amd.Engine(M, batch_sites=B)
N = sites + B
panel = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda")
eng.synth_device(panel.data_ptr(), 0, N, seed=0x1A2B3C, kind=int(os.environ.get("KIND", "0")))
eng.sync()
opts = amd.OPT_WITH_D | (amd.OPT_WITHIN_HIST if "h" in what else 0) | (amd.OPT_PACK3 if "p" in what else 0) | (amd.OPT_CHECKSUM if "c" in what else 0)
eng.pass_begin(N)
eng.pass_advance(panel.data_ptr(), B, B + 8, opts)
eng.sync()
t0 = time.perf_counter()
eng.pass_advance(panel.data_ptr() + B * eng.wpc * 4, sites, sites, opts)
eng.pass_end(opts)
dt = time.perf_counter() - t0
ms, n = eng.chain_timing()
print("M %d sites %d opts %s: %.3f us/site end to end, %.3e site*haps/s; chain %.2f us/launch" % (M, sites, what, 1e6 * dt / sites, M * sites / dt, 1e3 * ms / max(n, 1)))
