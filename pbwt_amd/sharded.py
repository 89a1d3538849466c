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



def _check_counts(send_counts, recv_counts, group=None):
    """PBWTAMD_SHARDED_CHECK=1 (debugging): all-gather every rank's send counts and compare them with the receive counts this
    rank derived locally — a disagreement would otherwise hang the all-to-all instead of raising"""
    import os
    if not os.environ.get("PBWTAMD_SHARDED_CHECK"):
        return
    import torch.distributed as dist
    allc = [None] * dist.get_world_size(group)
    dist.all_gather_object(allc, list(send_counts), group=group)
    me = dist.get_rank(group)
    got = [allc[s][me] for s in range(len(allc))]
    if got != list(recv_counts):
        raise RuntimeError("sharded exchange: derived receive counts %s != the senders' counts %s" % (list(recv_counts), got))


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
    _check_counts(send_counts, recv_counts, group)
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


# ---------------------------------------------------------------------------------------------------------------------
# Eight sites per exchange: the skeleton round with A RANK IN THE ROLE OF A TILE.
#
# a_{k+8} is the stable sort of a_k by the 8-bit key (bit j = the allele at site k+j), and every divergence of state k+8
# is a static function of (keys, d_k): an element whose key occurred before (anywhere in the order) gets the maximum of
# d_k over (predecessor, itself]; the first element of a key gets k + 1 + msb(key ^ key') with key' the nearest lower
# key present in the panel; new position 0 gets the sentinel.  What the three launches of the one-GPU round pass between
# tiles is exactly what the ranks exchange here:
#   hist  -> per rank and key (count, max d after the key's last occurrence | the rank's max d)   : ONE all-gather of 256 x 2
#   scan  -> every rank folds the rows of the ranks before it (skel_k2_kernel's combine), locally
#   rank  -> destination = bucket base + same-key elements on earlier ranks + local rank; (destination, a, d') : ONE all-to-all
# i.e. two collectives per EIGHT sites instead of two per site (sharded_step_AD), and the all-to-all's receive counts
# follow from the gathered rows.  Same tensors-stay-on-their-device rule: CPU tensors under gloo (tests/test_dist.py, every
# round against the oracle), device tensors under RCCL.  A protocol with torch kernels, not a tuned path: the natural
# device form scatters with peer stores from skel_rank_kernel and was not built without a multi-GPU box to run it on.

_MSB = None


def _msb_table(dev):
    global _MSB
    if _MSB is None or _MSB.device != dev:
        t = torch.zeros(256, dtype=torch.int64)
        for v in range(1, 256):
            t[v] = v.bit_length() - 1
        _MSB = t.to(dev)
    return _MSB


def _sparse_table(d):
    """levels[j][i] = max d[i : i + 2^j] (clipped at the end)"""
    levels = [d]
    n, w = d.numel(), 1
    while 2 * w <= n:
        prev = levels[-1]
        levels.append(torch.maximum(prev[: n - 2 * w + 1], prev[w: n - w + 1]))
        w *= 2
    return levels


def _range_max(levels, lo, hi):
    """max d[lo : hi + 1] elementwise for index tensors lo <= hi"""
    length = hi - lo + 1
    j = torch.floor(torch.log2(length.to(torch.float64))).to(torch.int64)
    j = torch.where((1 << j) > length, j - 1, j)                      # guard the float rounding at exact powers of two
    j = torch.where((1 << (j + 1)) <= length, j + 1, j)
    out = torch.zeros_like(lo)
    for lv, tab in enumerate(levels):
        sel = j == lv
        if bool(sel.any()):
            l, h = lo[sel], hi[sel]
            out[sel] = torch.maximum(tab[l], tab[h - (1 << lv) + 1])
    return out


def sharded_round8(a_loc, d_loc, key_loc, k, M, group=None):
    """eight sites of pbwtCursorForwardsAD (pbwtCore.c:485-508) on a position-sharded cursor, two collectives.
    a_loc, d_loc: this rank's positions of state k (int64); key_loc: their 8-bit keys (bit j = allele at site k + j).
    Returns (a_loc, d_loc) of state k + 8 for the same ownership ranges."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = a_loc.device
    bounds = owner_ranges(M, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n = a_loc.numel()
    keys = torch.arange(256, dtype=torch.int64, device=dev)
    # ---- "hist": this rank's row
    cnt = torch.bincount(key_loc, minlength=256).to(torch.int64)
    if n:
        pos = torch.arange(n, dtype=torch.int64, device=dev)
        last = torch.full((256,), -1, dtype=torch.int64, device=dev)
        last = last.scatter_reduce(0, key_loc, pos, reduce="amax", include_self=True)
        suf = torch.cat([torch.flip(torch.cummax(torch.flip(d_loc, [0]), 0).values, [0]), torch.zeros(1, dtype=torch.int64, device=dev)])
        tail = torch.where(cnt > 0, suf[torch.clamp(last, min=0) + 1], d_loc.max())
    else:
        tail = torch.zeros(256, dtype=torch.int64, device=dev)
    row = torch.stack([cnt, tail], dim=1).contiguous()
    rows = [torch.zeros_like(row) for _ in range(world)]
    dist.all_gather(rows, row, group=group)                            # collective 1 of 2
    allrows = torch.stack(rows)                                        # [world][256][2]
    # ---- "scan": fold the ranks before this one; bucket bases; nearest lower key present
    before = torch.zeros(256, dtype=torch.int64, device=dev)
    carry = torch.full((256,), -1, dtype=torch.int64, device=dev)
    for r in range(rank):
        c, t = allrows[r, :, 0], allrows[r, :, 1]
        carry = torch.where(c > 0, t, torch.where(carry >= 0, torch.maximum(carry, t), carry))
        before = before + c
    total = allrows[:, :, 0].sum(0)
    G = torch.cumsum(total, 0) - total
    present = torch.where(total > 0, keys, torch.full_like(keys, -1))
    lower = torch.cat([torch.full((1,), -1, dtype=torch.int64, device=dev), torch.cummax(present, 0).values[:-1]])
    # ---- "rank": local stable order by key, predecessors, range maxima, destinations
    if n:
        order = torch.sort(key_loc, stable=True).indices              # positions grouped by key, ascending inside a key
        ks = key_loc[order]
        first = torch.ones(n, dtype=torch.bool, device=dev)
        first[1:] = ks[1:] != ks[:-1]
        idx = torch.arange(n, dtype=torch.int64, device=dev)
        start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
        rk = idx - start                                               # rank among the same key on this rank
        pred = torch.where(first, torch.full_like(order, -1), torch.cat([order[:1], order[:-1]]))
        levels = _sparse_table(d_loc)
        rm = _range_max(levels, torch.where(pred >= 0, pred + 1, torch.zeros_like(pred)), order)
        cy = carry[ks]
        lw = lower[ks]
        dd_first = torch.where(cy >= 0, torch.maximum(cy, rm),
                               torch.where(lw >= 0, k + 1 + _msb_table(dev)[ks ^ torch.clamp(lw, min=0)], torch.zeros_like(rm)))
        dd = torch.where(pred >= 0, rm, dd_first)
        dest = G[ks] + before[ks] + rk
        dd = torch.where(dest == 0, torch.full_like(dd, k + 9), dd)    # sentinel of state k + 8 (pbwtCore.c:507 after the eighth site)
        a_s = a_loc[order]
    else:
        dest = torch.zeros(0, dtype=torch.int64, device=dev); dd = dest; a_s = dest
    # ---- all-to-all of (destination, a, d'), grouped by the owner of the destination
    edges = torch.tensor(bounds[1:], dtype=torch.int64, device=dev)
    owner = torch.bucketize(dest, edges, right=True)
    o2 = torch.sort(owner, stable=True).indices
    send = torch.stack([dest[o2], a_s[o2], dd[o2]], dim=1).reshape(-1).contiguous()
    send_counts = torch.bincount(owner, minlength=world).cpu().tolist()
    # receive counts from the gathered rows: rank s sends its elements of key x to [G[x] + before_s[x], ... + count_s[x])
    cs = allrows[:, :, 0]
    bef_all = torch.cumsum(cs, 0) - cs                                 # [world][256]: same-key elements on earlier ranks
    b0 = G.unsqueeze(0) + bef_all
    ov = torch.clamp(torch.minimum(b0 + cs, torch.full_like(b0, hi)) - torch.maximum(b0, torch.full_like(b0, lo)), min=0)
    recv_counts = ov.sum(1).cpu().tolist()
    _check_counts(send_counts, recv_counts, group)
    recv = torch.zeros(3 * sum(recv_counts), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[3 * c for c in recv_counts],
                           input_split_sizes=[3 * c for c in send_counts], group=group)     # collective 2 of 2
    rv = recv.reshape(-1, 3)
    a_new = torch.empty(n, dtype=torch.int64, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    a_new[rv[:, 0] - lo] = rv[:, 1]
    d_new[rv[:, 0] - lo] = rv[:, 2]
    return a_new, d_new
