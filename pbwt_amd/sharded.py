"""Position-sharded PBWT step over torch.distributed — the exchange protocol of SURVEY.md §8e(1).

Rank r owns the contiguous positions [lo_r, hi_r) of the current order (a_k, d_k).  One site costs
  1. a local fold of the rank's segment into the carry tuple (c0, c1, t0, t1, all) — the same
     associative operator the HIP kernels use between tiles (pbwt_kernels.h: tup_combine);
  2. ONE all-gather of those 5 ints per rank: every rank derives its zero offset, the column total,
     its incoming running maxima p, q (pbwtCore.c:492-503) and everybody's destination ranges;
  3. the local step, producing (destination, a, d') per owned position;
  4. ONE all-to-all of (destination, a, d') triples: a rank's zeros go to one contiguous destination
     range and its ones to another, so it talks to at most a few peers; receive counts follow from
     the gathered tuples, no size exchange is needed.
This module is the protocol, written with numpy on host tensors so it runs under gloo (tests) and,
unchanged, under RCCL.  It is NOT the default multi-GPU mode: one RCCL collective costs 10-20 us
while a whole site costs 4.7 us (M = 100k) to 14 us (M = 1M) on a single MI355X, so
sharding positions across GPUs slows the recurrence down until M >> 10M.  bench.py --gpus N runs
independent panels per rank instead.
"""
import numpy as np
import torch
import torch.distributed as dist


def owner_ranges(M, world):
    per, extra = divmod(M, world)
    lo = [r * per + min(r, extra) for r in range(world)]
    return lo + [M]


def tup_of(y, d):
    """carry tuple of a segment: c0, c1, t_b = max d after the last allele-b element (all if none), all"""
    c0 = int((y == 0).sum()); c1 = int(len(y) - c0)
    allm = int(d.max()) if len(d) else 0
    def tail(b):
        idx = np.nonzero(y == b)[0]
        if len(idx) == 0:
            return allm
        after = d[idx[-1] + 1:]
        return int(after.max()) if len(after) else 0
    return np.array([c0, c1, tail(0), tail(1), allm], dtype=np.int64)


def tup_combine(L, R):
    out = np.empty(5, dtype=np.int64)
    out[0] = L[0] + R[0]; out[1] = L[1] + R[1]
    out[4] = max(L[4], R[4])
    out[2] = R[2] if R[0] else max(L[2], R[4])
    out[3] = R[3] if R[1] else max(L[3], R[4])
    return out


def sharded_step_AD(a_loc, d_loc, y_loc, k, M, group=None):
    """one site of pbwtCursorForwardsAD on a position-sharded cursor.
    a_loc, y_loc: this rank's positions; d_loc: d at those positions (d[M] is handled by the last rank).
    Returns (a_new_loc, d_new_loc) for the same ownership ranges of the new order."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = owner_ranges(M, world)
    lo = bounds[rank]
    n = len(a_loc)
    # 1-2: all-gather of the carry tuples
    mine = torch.from_numpy(tup_of(y_loc, d_loc))
    gathered = [torch.zeros(5, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    tups = [g.numpy() for g in gathered]
    C = sum(int(t[0]) for t in tups)                               # zeros in the column (u->c)
    pre = np.array([0, 0, k + 1, k + 1, k + 1], dtype=np.int64)     # p = q = k+1 before position 0 (pbwtCore.c:489)
    pre_seen = [False, False]
    zeros_before = 0
    for r in range(rank):
        pre = tup_combine(pre, tups[r]); zeros_before += int(tups[r][0])
    p, q = int(pre[2]), int(pre[3])
    # 3: local step
    dest = np.empty(n, dtype=np.int64); dnew = np.empty(n, dtype=np.int64)
    zi, oi = zeros_before, C + (lo - zeros_before)
    for i in range(n):
        di = int(d_loc[i])
        p = max(p, di); q = max(q, di)
        if y_loc[i] == 0:
            dest[i] = zi; dnew[i] = p; zi += 1; p = 0
        else:
            dest[i] = oi; dnew[i] = q; oi += 1; q = 0
    # 4: all-to-all of (destination, a, d') triples, grouped by owner of the destination
    owner = np.searchsorted(np.array(bounds[1:]), dest, side="right")
    order = np.argsort(owner, kind="stable")
    send = torch.from_numpy(np.stack([dest[order], a_loc[order].astype(np.int64), dnew[order]], axis=1).reshape(-1).copy())
    send_counts = [int((owner == r).sum()) for r in range(world)]
    # receive counts from the gathered tuples alone: rank s sends its zeros to [Zs, Zs+c0_s) and its ones to [C+Os, ...)
    recv_counts = []
    zb = 0
    for s in range(world):
        c0s, c1s = int(tups[s][0]), int(tups[s][1])
        ob = bounds[s] - zb
        def overlap(a0, a1):
            return max(0, min(a1, bounds[rank + 1]) - max(a0, bounds[rank]))
        recv_counts.append(overlap(zb, zb + c0s) + overlap(C + ob, C + ob + c1s))
        zb += c0s
    recv = torch.zeros(3 * sum(recv_counts), dtype=torch.int64)
    dist.all_to_all_single(recv, send, output_split_sizes=[3 * c for c in recv_counts],
                           input_split_sizes=[3 * c for c in send_counts], group=group)
    rv = recv.numpy().reshape(-1, 3)
    a_new = np.empty(n, dtype=a_loc.dtype); d_new = np.empty(n, dtype=np.int64)
    a_new[rv[:, 0] - lo] = rv[:, 1]; d_new[rv[:, 0] - lo] = rv[:, 2]
    if rank == 0 and n:
        d_new[0] = k + 2                                           # sentinel (pbwtCore.c:507); d[M] = k+2 is implicit
    return a_new, d_new
