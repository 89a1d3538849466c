"""Position-sharded PBWT step over torch.distributed — the exchange protocol of SURVEY.md §8e(1) / the north star's
"per-site exclusive scan of local 0/1 counts and an all-to-all".

Rank r owns the contiguous positions [lo_r, hi_r) of the current order (a_k, d_k) as torch tensors on ITS device.
One site costs
  1. a local fold of the rank's segment into the carry tuple (c0, c1, t0, t1, all) — the same associative operator
     the HIP kernels use between tiles (pbwt_kernels.h: tup_combine);
  2. ONE all-gather of those 5 integers per rank: every rank derives its zero offset, the column total, its incoming
     running maxima p, q (pbwtCore.c:492-503) and everybody's destination ranges;
  3. the local step as tensor operations (segmented running maxima through one cummax per allele), producing
     (destination, a, d') per owned position;
  4. ONE all-to-all of (destination, a, d') triples: a rank's zeros go to one contiguous destination range and its
     ones to another, so it talks to at most a few peers; the receive counts follow from the gathered tuples, no size
     exchange is needed.
Every tensor stays on the device it was given on and the two collectives are plain torch.distributed calls, so the same
function runs with CPU tensors under gloo (tests/test_dist.py, world_size 2 and 3, every site against the oracle) and
with device tensors under RCCL ("nccl").  It has NOT been timed on a multi-GPU node (none was available to this
build); by the measured cost of a collective (10-20 us) against a whole site on one MI355X (1.7 us at M = 100 k, 5-10
us at 1 M) it is expected to lose to a single GPU until M >> 10 M, which is why bench.py's multi-GPU modes are
independent panels per rank (default) and site-block sharding of one panel (pbwt_amd/siteblock.py), not this.
"""
import torch
import torch.distributed as dist

BIG = 1 << 32          # divergences are < 2^31: one segment's values never reach the next segment's offset


def owner_ranges(M, world):
    per, extra = divmod(M, world)
    lo = [r * per + min(r, extra) for r in range(world)]
    return lo + [M]


def tup_of(y, d):
    """carry tuple of a segment (int64 tensor of 5): c0, c1, t_b = max d after the last allele-b element (all if
    there is none), all"""
    n = y.numel()
    if n == 0:
        return torch.zeros(5, dtype=torch.int64, device=y.device)
    pos = torch.arange(n, device=y.device)
    is0 = y == 0
    c0 = is0.sum()
    allm = d.max()
    # suffix maxima: sm[i] = max d[i:]  (sm[n] = 0)
    sm = torch.cat([torch.flip(torch.cummax(torch.flip(d, [0]), 0).values, [0]), torch.zeros(1, dtype=d.dtype, device=d.device)])
    last0 = torch.where(is0, pos, torch.full_like(pos, -1)).max()
    last1 = torch.where(~is0, pos, torch.full_like(pos, -1)).max()
    t0 = torch.where(last0 >= 0, sm[last0 + 1], allm)
    t1 = torch.where(last1 >= 0, sm[last1 + 1], allm)
    return torch.stack([c0, n - c0, t0, t1, allm]).to(torch.int64)


def tup_combine(L, R):
    """python ints: the combine of pbwt_kernels.h::tup_combine"""
    return [L[0] + R[0], L[1] + R[1],
            R[2] if R[0] else max(L[2], R[4]),
            R[3] if R[1] else max(L[3], R[4]),
            max(L[4], R[4])]


def _segmented_running_max(d, seg):
    """m[i] = max of d over the elements j <= i with seg[j] == seg[i]; seg is non-decreasing"""
    v = d + seg * BIG
    return torch.cummax(v, 0).values - seg * BIG


def sharded_step_AD(a_loc, d_loc, y_loc, k, M, group=None):
    """one site of pbwtCursorForwardsAD (pbwtCore.c:485-508) on a position-sharded cursor.
    a_loc, d_loc, y_loc: int64 tensors of this rank's positions (d at those positions; d[M] is implicit).
    Returns (a_new_loc, d_new_loc) for the same ownership ranges of the new order, on the same device."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = a_loc.device
    bounds = owner_ranges(M, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n = a_loc.numel()
    # 1-2: all-gather of the carry tuples
    mine = tup_of(y_loc, d_loc)
    gathered = [torch.zeros(5, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    tups = torch.stack(gathered).cpu().tolist()                     # world x 5 python ints (the only host round trip)
    C = sum(t[0] for t in tups)                                     # zeros in the column (u->c)
    pre = [0, 0, k + 1, k + 1, k + 1]                               # p = q = k+1 before position 0 (pbwtCore.c:489)
    zeros_before = 0
    for r in range(rank):
        pre = tup_combine(pre, tups[r]); zeros_before += tups[r][0]
    p_in, q_in = pre[2], pre[3]
    # 3: local step.  p before element i = max d since the last zero (exclusive) -> segments delimited by the zeros
    is0 = y_loc == 0
    z_excl = torch.cumsum(is0.to(torch.int64), 0) - is0.to(torch.int64)          # zeros strictly before i
    o_excl = torch.cumsum((~is0).to(torch.int64), 0) - (~is0).to(torch.int64)          # ones strictly before i
    m0 = _segmented_running_max(d_loc, z_excl)                       # running max inside the zero-delimited segment
    m1 = _segmented_running_max(d_loc, o_excl)
    m0 = torch.where(z_excl == 0, torch.clamp(m0, min=p_in), m0)     # the first segment continues the previous ranks' run
    m1 = torch.where(o_excl == 0, torch.clamp(m1, min=q_in), m1)
    dnew = torch.where(is0, m0, m1)
    dest = torch.where(is0, zeros_before + z_excl, C + (lo - zeros_before) + o_excl)
    # 4: all-to-all of (destination, a, d') triples, grouped by the owner of the destination
    edges = torch.tensor(bounds[1:], dtype=torch.int64, device=dev)
    owner = torch.bucketize(dest, edges, right=True)
    order = torch.sort(owner, stable=True).indices
    send = torch.stack([dest[order], a_loc[order], dnew[order]], dim=1).reshape(-1).contiguous()
    send_counts = torch.bincount(owner, minlength=world).cpu().tolist()
    # receive counts from the gathered tuples alone: rank s sends its zeros to [Zs, Zs+c0_s) and its ones to [C+Os, ...)
    recv_counts = []
    zb = 0
    for s in range(world):
        c0s, c1s = tups[s][0], tups[s][1]
        ob = bounds[s] - zb

        def overlap(a0, a1):
            return max(0, min(a1, hi) - max(a0, lo))
        recv_counts.append(overlap(zb, zb + c0s) + overlap(C + ob, C + ob + c1s))
        zb += c0s
    recv = torch.zeros(3 * sum(recv_counts), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[3 * c for c in recv_counts],
                           input_split_sizes=[3 * c for c in send_counts], group=group)
    rv = recv.reshape(-1, 3)
    a_new = torch.empty(n, dtype=torch.int64, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    a_new[rv[:, 0] - lo] = rv[:, 1]
    d_new[rv[:, 0] - lo] = rv[:, 2]
    if rank == 0 and n:
        d_new[0] = k + 2                                            # sentinel (pbwtCore.c:507); d[M] = k+2 is implicit
    return a_new, d_new
