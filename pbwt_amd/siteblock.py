"""Site-block sharding of ONE panel across the GPUs of a node — the device path of row (e).

The PBWT recurrence is sequential in the sites, but the expensive part of `build + -maxWithin` is not the
recurrence: per site the dependent chain costs t_c (skeleton rounds only: 1.45 us at M = 100 k, 5 us at 1 M) while
the full hot path — chain + the seven fill states + the matchMaximalWithin sweep + pack3 — costs t_f (1.7 / 10.5 us).
So the sites are cut into G consecutive blocks, one per rank, and rank g

  1. runs the CHAIN ONLY (pbwtamd_pass_advance with OPT_WITH_D and no consumer) over the sites before its block: the
     "pass 1 ... every GPU runs pass 1 redundantly: zero communication" variant of SURVEY §8e(2) — the (a, d)
     checkpoint at its block start is computed where it is needed, nothing crosses xGMI on the data path;
  2. runs the full hot path over its own block [k_lo, k_hi) (the last rank also the closing k == N sweep);
  3. stops (pbwtamd_pass_stop).

Every rank finishes at the same time when the blocks shrink geometrically, b_g = b_0 (1 - rho)^g with
rho = t_c / t_f (`plan_blocks`): wall time N t_c / (1 - (1 - rho)^G) instead of N t_f — 1.5x / 1.9x / 2.1x at
G = 2 / 4 / 8 for M = 1 M (rho = 0.48), bounded by t_f / t_c because the recurrence itself stays serial.

What does cross the wire, once, after the timed work: the -stats histogram (one all-reduce SUM — RCCL under the
"nccl" backend, gloo in the CPU tests) and, if wanted, the ranks' pack3 byte blocks, which concatenate in rank order
into the panel's .pbwt payload (runs never span columns, pbwtCore.c:254-267).  The same functions run under gloo
with CPU tensors (tests/test_dist.py) and under RCCL with device tensors (bench.py --mode siteblock).
A checkpointed cursor can also be shipped instead of recomputed: pbwtamd_pass_begin(a_k, k0) + pbwtamd_pass_set_d(d_k)
restart a pass exactly where another one stopped (tests/test_gpu_multi.py::test_restart_from_checkpoint)."""
import numpy as np

# measured t_chain / t_full per site on one MI355X, used when the caller passes no rho
RHO_BY_M = ((150000, 0.84), (400000, 0.65), (1 << 62, 0.48))


def default_rho(M):
    for lim, r in RHO_BY_M:
        if M <= lim:
            return r
    return 0.5


def plan_blocks(N, world, rho, align=512):
    """G consecutive site blocks [k_lo, k_hi) covering [0, N), every boundary a multiple of `align` (the engine's
    batch, so that every batch inside a block is a full skeleton batch), sized b_g ~ b_0 (1 - rho)^g so that
    prefix(g) * rho + b_g is the same for every rank.  Later ranks get the smaller blocks."""
    if world < 1 or N < 0:
        raise ValueError("plan_blocks: world %d, N %d" % (world, N))
    rho = min(max(float(rho), 0.0), 0.999)
    w = np.array([(1.0 - rho) ** g for g in range(world)], dtype=np.float64)
    edges = np.concatenate([[0.0], np.cumsum(w / w.sum())]) * N
    ks = [0]
    for g in range(1, world):
        k = int(round(edges[g] / align)) * align
        ks.append(min(max(k, ks[-1]), (N // align) * align))
    ks.append(N)
    return [(ks[g], ks[g + 1]) for g in range(world)]


def model_time(blocks, rho):
    """per-rank time in units of N * t_f: the chain-only prefix at rho per site, the block at 1"""
    return [lo * rho + (hi - lo) for lo, hi in blocks]


def run_block(eng, col_ptr, N, block, opts, is_last, step=8192):
    """this rank's share of the panel on its GPU: chain-only prefix, then the block with the consumers in `opts`.
    col_ptr(k) -> device address of bit column k (the whole panel, or at least [0, k_hi + 8), is resident)."""
    from . import api
    k_lo, k_hi = block
    eng.pass_begin(N)
    chain_only = opts & (api.OPT_WITH_D | api.OPT_SORTED) | api.OPT_WITH_D
    for lo, hi, o in ((0, k_lo, chain_only), (k_lo, k_hi, opts)):
        k = lo
        while k < hi:
            n = min(step, hi - k)
            eng.pass_advance(col_ptr(k), n, min(n + 8, N - k), o)
            k += n
    if is_last and k_hi == N:
        eng.pass_end(opts)                      # the closing k == N sweep belongs to the block that ends the panel
    else:
        eng.pass_stop()


def reduce_hist(hist, device=None):
    """sum of the ranks' -stats histograms (int64 numpy in, int64 numpy out on every rank): ONE all-reduce"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return hist
    t = torch.from_numpy(np.ascontiguousarray(hist, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_packed(yz_block, dst=0):
    """the ranks' pack3 byte blocks -> the panel's yz on rank `dst` (None elsewhere), in site order = rank order"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return yz_block
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(np.ascontiguousarray(yz_block, dtype=np.uint8).tobytes(), parts, dst=dst)
    if parts is None:
        return None
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def calibrate_rho(eng, col_ptr, N, opts, batch, nbatches=4):
    """t_chain / t_full measured on this GPU with this panel: a few batches chain-only, then the same with consumers"""
    import time
    from . import api
    n = min(N, batch * nbatches)
    if n < batch:
        return default_rho(eng.M)
    t = []
    for o in (api.OPT_WITH_D, opts):
        eng.pass_begin(N)
        eng.pass_advance(col_ptr(0), batch, min(batch + 8, N), o)     # warm-up batch
        eng.sync()
        t0 = time.perf_counter()
        if n > batch:
            eng.pass_advance(col_ptr(batch), n - batch, min(n - batch + 8, N - batch), o)
        eng.pass_stop()
        t.append(time.perf_counter() - t0)
    return min(max(t[0] / max(t[1], 1e-9), 0.05), 0.99)
