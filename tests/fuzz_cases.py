"""randomised differential cases: every device path against the oracle (shared by tests/test_gpu_fuzz.py, which runs a
fixed seed budget under `-m gpu`, and tools/fuzz_gpu.py, the open-ended command-line run)"""
import time

import numpy as np


def run_cases(seed, ncases=None, budget_s=None, verbose=False):
    """returns (cases run, list of mismatch descriptions)"""
    import torch
    import pbwt_amd as amd
    import oracle as orc
    rng = np.random.default_rng(seed)
    t0 = time.time(); n = 0; bad = []
    while (ncases is None or n < ncases) and (budget_s is None or time.time() - t0 < budget_s):
        M = int(rng.choice([2, 3, 17, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 1024, 1025, 2047, 3000, 5000, 40000, 40001, 70000]))
        N = int(rng.integers(1, 200)) if M < 10000 else int(rng.integers(8, 60))
        B = int(rng.choice([2, 8, 16, 24, 40, 64, 128, 512]))
        kind = int(rng.integers(0, 2))
        bits = orc.synth_bitcols(M, N, seed=int(rng.integers(1, 1 << 30)), kind=kind)
        o = orc.build_bitcols(bits, M, with_d=True)
        eng = amd.Engine(M, batch_sites=B)
        mode = int(rng.integers(0, 6))
        ok = True
        try:
            if mode == 0:      # device pass API, every site, split into random advances
                buf = torch.from_numpy(bits.view(np.int32)).cuda()
                opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST | (amd.OPT_PACK3 if rng.random() < 0.5 else 0)
                eng.pass_begin(N)
                k = 0
                while k < N:
                    step = int(min(N - k, rng.choice([1, 7, 8, 16, 33, 64, 1000])))
                    avail = int(min(N - k, step + rng.choice([1, 2, 8, 9])))
                    eng.pass_advance(buf.data_ptr() + k * eng.wpc * 4, step, avail, opts); k += step
                eng.pass_end(opts)
                a, d = eng.get_state(); ca, cd, _ = eng.get_checksums(0, N + 1)
                ok = np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"]) and np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
                ok = ok and np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
                if opts & amd.OPT_PACK3: ok = ok and np.array_equal(eng.get_packed(), o["yz"])
            elif mode == 1:    # packed consumers (no ids)
                buf = torch.from_numpy(bits.view(np.int32)).cuda()
                opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
                eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
                ok = np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1]) and np.array_equal(eng.get_packed(), o["yz"])
            elif mode == 5:    # packed consumers checked at EVERY position: checksums of d and y taken from the hand-off slots (16-bit with a random escape threshold, or 32-bit)
                import os
                saved = {k: os.environ.get(k) for k in ("PBWTAMD_PACKED_CHECKSUM", "PBWTAMD_P16", "PBWTAMD_P16_CLIP")}
                os.environ["PBWTAMD_PACKED_CHECKSUM"] = "1"
                os.environ["PBWTAMD_P16"] = "1" if rng.random() < 0.8 else "0"
                os.environ["PBWTAMD_P16_CLIP"] = str(int(rng.choice([1, 2, 5, 37, 32767])))
                try:
                    buf = torch.from_numpy(bits.view(np.int32)).cuda()
                    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3 | amd.OPT_CHECKSUM
                    eng.pass_begin(N)
                    k = 0
                    while k < N:
                        step = int(min(N - k, rng.choice([8, 16, 64, 1000])))
                        eng.pass_advance(buf.data_ptr() + k * eng.wpc * 4, step, int(min(N - k, step + 8)), opts); k += step
                    eng.pass_end(opts)
                    _, cd, cy = eng.get_checksums(0, N)
                    sw = orc.sweep_AD(o["yz"], M, N)
                    ok = np.array_equal(cd[:N], o["csum_d"][:N]) and np.array_equal(cy[:N], sw["csum_y"][:N])
                    ok = ok and np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1]) and np.array_equal(eng.get_packed(), o["yz"])
                finally:
                    for k2, v in saved.items():
                        if v is None: os.environ.pop(k2, None)
                        else: os.environ[k2] = v
            elif mode == 2:    # host build + read side
                b = eng.build(bits, with_d=bool(rng.integers(0, 2)))
                ok = np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"])
                sw = eng.sweep_AD(o["yz"], N); s = orc.sweep_AD(o["yz"], M, N)
                ok = ok and np.array_equal(sw["csum_a"], s["csum_a"]) and np.array_equal(sw["csum_d"], s["csum_d"]) and np.array_equal(sw["csum_y"][:N], s["csum_y"][:N])
            elif mode == 4:    # query sweeps (dense and sparse)
                if M <= 5000 and M >= 4:
                    Mq = int(rng.integers(1, min(M - 1, 40)))
                    hap = orc.unpack_bitcols(bits, M)
                    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :M - Mq]), M - Mq, with_d=False)["yz"]
                    qz = orc.build_bitcols(orc.pack_bitcols(hap[:, M - Mq:]), Mq, with_d=False)["yz"]
                    e2 = amd.Engine(M - Mq, batch_sites=max(B, 8))
                    got, gn, gt = e2.match_sweep(pz, N, qz, Mq)
                    want, wn, wt = orc.match_sweep(pz, M - Mq, qz, Mq, N)
                    ok = np.array_equal(got, want) and gn == wn and tuple(gt) == tuple(wt)
                    nS = int(rng.integers(1, 6))
                    got, gn, gt = e2.match_sweep_sparse(pz, N, qz, Mq, nS)
                    want, wn, wt = orc.match_sweep_sparse(pz, M - Mq, qz, Mq, N, nS)
                    ok = ok and np.array_equal(got, want) and gn == wn and tuple(gt) == tuple(wt)
                    e2.close()
            else:              # records
                if M <= 3000:
                    ok = np.array_equal(eng.max_within(o["yz"], N, mode="records"), orc.max_within(o["yz"], M, N))
                    L = int(rng.integers(1, 30))
                    ok = ok and np.array_equal(eng.long_within(o["yz"], N, L), orc.long_within(o["yz"], M, N, L))
        except Exception as ex:
            ok = False; bad.append("EXC %r" % (ex,))
        n += 1
        if not ok:
            bad.append("MISMATCH M=%d N=%d B=%d kind=%d mode=%d (case %d of seed %d)" % (M, N, B, kind, mode, n, seed))
            if verbose: print(bad[-1])
        eng.close()
    return n, bad
