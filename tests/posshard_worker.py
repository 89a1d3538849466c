"""Run under torch.distributed.run by tests/test_gpu_multi.py (and by hand on the GPU box): `world` ranks on the box's ONE GPU,
each an engine of the same panel in position-sharded mode (pbwt_amd/posshard.py, csrc/pbwt_shard.inc); gloo carries the
handle blobs and the once-per-job gathers.  Rank 0 compares with the oracle: every site's a/d/y checksum, the summed
-stats histogram, the interleaved pack3 bytes, the final a and d."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pbwt_amd import dist as pd
    from pbwt_amd import posshard as ps
    import pbwt_amd as amd
    rank, world = pd.init("gloo")
    M, N, B = int(os.environ["PS_M"]), int(os.environ["PS_N"]), int(os.environ.get("PS_B", "512"))
    kind, step = int(os.environ.get("PS_KIND", "0")), int(os.environ.get("PS_STEP", "8192"))
    with_csum = int(os.environ.get("PS_CSUM", "1"))
    trace = (lambda m: print("[worker %d] %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("PS_TRACE") else (lambda m: None)
    # PS_SPREAD=1 (tests/test_gpu_multi.py::test_position_sharded_across_devices, multi-GPU nodes only): rank r on device r mod device_count — the
    # peer stores, the flag barriers and the hipIpc mappings then really cross xGMI; default: every rank on device 0
    dev = (rank % max(torch.cuda.device_count(), 1)) if os.environ.get("PS_SPREAD") == "1" else 0
    torch.cuda.set_device(dev)
    eng = amd.Engine(M, batch_sites=B, device=dev)
    trace("engine up on device %d" % dev)
    ps.setup(eng, rank, world)
    trace("shard connected")
    panel = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda:%d" % dev)
    torch.cuda.synchronize()
    eng.synth_device(panel.data_ptr(), 0, N, seed=0x9051, kind=kind)       # every rank holds the same columns
    eng.sync()
    trace("panel resident")
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3 | (amd.OPT_CHECKSUM if with_csum else 0)
    pd.barrier()
    t0 = time.perf_counter()
    ps.run(eng, lambda k: panel.data_ptr() + k * eng.wpc * 4, N, opts, step=step)
    dt = time.perf_counter() - t0
    trace("pass done in %.3f s" % dt)
    hist = ps.reduce_hist(eng.get_hist(N + 1))
    cs = ps.gather_checksums(eng, 0, N + 1) if with_csum else None
    yz = ps.gather_packed(eng)
    a, d = eng.get_state()
    trace("results gathered")
    res = {"rank": rank, "device": dev, "range": list(eng.shard_range(rank)), "seconds": dt, "blocks": int(len(eng.shard_blocks()[0]))}
    if rank == 0:
        import oracle
        bits = panel.cpu().numpy().view(np.uint32)
        o = oracle.build_bitcols(bits, M, with_d=True, want_csum=bool(with_csum))
        res["yz"] = bool(np.array_equal(yz, o["yz"]))
        res["hist"] = bool(np.array_equal(hist, oracle.max_within_hist(o["yz"], M, N)[: N + 1]))
        if with_csum:
            res["csum_a"] = bool(np.array_equal(cs[0], o["csum_a"])); res["csum_d"] = bool(np.array_equal(cs[1], o["csum_d"]))
            if not res["csum_a"]:
                res["first_bad_a"] = int(np.nonzero(cs[0] != o["csum_a"])[0][0])
            if not res["csum_d"]:
                res["first_bad_d"] = int(np.nonzero(cs[1] != o["csum_d"])[0][0])
        res["aFend"] = bool(np.array_equal(a, o["aFend"])); res["dFend"] = bool(np.array_equal(d, o["d_final"]))
        res["ok"] = all(v for k2, v in res.items() if isinstance(v, bool))
    else:                                                    # every rank ends with the complete final state: compare with rank 0's
        res["ok"] = True
    import torch.distributed as dist
    fa = [None] * world
    if world > 1:
        dist.all_gather_object(fa, (a.tobytes(), d.tobytes()))
    else:
        fa = [(a.tobytes(), d.tobytes())]
    res["same_final_state"] = all(x == fa[0] for x in fa)
    res["ok"] = bool(res["ok"] and res["same_final_state"])
    with open(os.path.join(os.environ["OUT_DIR"], "ps%d.json" % rank), "w") as f:
        json.dump(res, f)
    eng.close()
    pd.finish()


if __name__ == "__main__":
    main()
