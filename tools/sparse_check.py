import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pbwt_amd as amd, oracle as orc
for (Mp, Mq, N, kind, nS, B) in [(8, 3, 10, 1, 2, 4), (50, 7, 60, 1, 2, 16), (50, 7, 61, 0, 3, 16), (200, 20, 150, 0, 4, 32), (64, 10, 33, 1, 5, 10), (300, 25, 100, 0, 2, 512), (40, 6, 50, 1, 1, 16), (17, 4, 9, 1, 3, 8), (3000, 50, 200, 0, 4, 64)]:
    bits = orc.synth_bitcols(Mp + Mq, N, seed=Mp * 7 + N, kind=kind)
    hap = orc.unpack_bitcols(bits, Mp + Mq)
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]; qz = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    o, nom, tot = orc.match_sweep_sparse(pz, Mp, qz, Mq, N, nS)
    eng = amd.Engine(Mp, batch_sites=B)
    g, gn, gt = eng.match_sweep_sparse(pz, N, qz, Mq, nS)
    ok = len(o) == len(g) and np.array_equal(o, g) and nom == gn and tuple(tot) == tuple(gt)
    print(Mp, Mq, N, kind, nS, B, "records", len(o), len(g), "tot", tot, gt, "OK" if ok else "MISMATCH")
    if not ok and len(o) == len(g):
        bad = np.nonzero(o != g)[0][:5]; print("  first diffs", bad, o[bad], g[bad])
