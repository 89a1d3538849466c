"""-matchDynamic across the GPUs of a node by sharding the QUERIES ("for matchDynamic shard queries
instead: queries are independent given the panel state").

matchSequencesSweep (pbwtMatch.c:363-443) carries per query jj only f[jj] (first panel position of its current longest
match) and d[jj] (that match's start); the panel cursor and the query panel's own cursor are shared, read-only inputs of
the per-query update (`:376-414`).  So rank g of G

  1. runs the panel chain and the query-panel chain like a single GPU would (both are needed in full: the query panel's
     order uq->a decides the order of the reports),
  2. sweeps only the queries lo_g <= jj < hi_g (`pbwtamd_set_query_range`; the other waves exit at once), and
  3. tags every record with the query's rank in uq->a at the record's site.

(end site, rank, isSparse) is the reference's emission order — k ascending, then uq->a order, dense matches before sparse
ones (`pbwtMatch.c:375-385, 452-499`); the tails at N come cursor by cursor (`:577-594`): dense for every query in final
order, then each sparse cursor kk, which the reported start = nSparse * d + kk identifies.  `merge_streams` therefore
rebuilds EXACTLY the reference's stream from the per-rank streams with one stable sort; nTot / totLen / no-match counts
add up.  What crosses the wire, once, after the sweep: the records (20 B each) and three counters — an all-gather over
RCCL (backend "nccl") or gloo (CPU tests, tests/test_dist.py).

Speed-up bound, measured (bench.py `match_dynamic.query_sharding_one_rank_share`: one rank's share timed on one MI355X, M = 1 M,
10 000 queries, 8 192 sites): 17.9 us/site for all queries, 15.5 / 15.1 / 13.1 us/site for 1/2, 1/4, 1/8 of them — the
panel side (read-side chain with d, the full (a, d) fill, sorted columns and rank directories: ~12 us/site) is repeated by
every rank and only the per-query walks divide: 1.15x / 1.19x / 1.37x at G = 2 / 4 / 8.  Sharding the panel's sites instead
is not possible here (f[jj], d[jj] run through all sites); the lever is the panel side itself.  The mode
is exact and cheap to use, not a scaling result."""
import numpy as np

MERGED_DTYPE = np.dtype([("ai", "<i4"), ("bi", "<i4"), ("start", "<i4"), ("end", "<i4"), ("sparse", "<i4")])


def plan_ranges(Mq, world):
    """contiguous ranges [lo, hi) of original query indices, one per rank, sizes differing by at most one"""
    if world < 1 or Mq < 0:
        raise ValueError("plan_ranges: Mq %d, world %d" % (Mq, world))
    base, extra = divmod(Mq, world)
    out, lo = [], 0
    for g in range(world):
        hi = lo + base + (1 if g < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def run_range(eng, pz, N, qz, Mq, lo, hi, nSparse=0, pStart=None, qStart=None):
    """this rank's share on its GPU: records tagged with the query's rank (sparse >> 1), its no-match events
    (jj, x, k, isSparse | rank << 1), its no-match count and (nTot, totLen)"""
    eng.set_query_range(lo, hi)
    try:
        recs, nom, tot = eng.match_sweep_sparse(pz, N, qz, Mq, nSparse, pStart=pStart, qStart=qStart)
        ev = eng.nomatch_events() if nom else np.zeros((0, 4), np.int32)
    finally:
        eng.set_query_range(-1)
    return recs, ev, nom, tot


def merge_streams(streams, N, nSparse=0):
    """records of all ranks (each in its own emission order, rank-tagged) -> the reference's stream, tags stripped"""
    parts = [np.asarray(s) for s in streams if len(s)]
    if not parts:
        return np.zeros(0, MERGED_DTYPE)
    r = np.concatenate(parts)
    nS = nSparse if nSparse > 1 else 0
    end = r["end"].astype(np.int64)
    qrank = (r["sparse"] >> 1).astype(np.int64)
    sp = (r["sparse"] & 1).astype(np.int64)
    tail = end == N
    # sites k < N: (k, rank, dense before sparse); tails: (N, cursor, rank) with cursor 0 = dense, 1 + kk = sparse cursor kk
    cursor = np.where(sp == 1, 1 + (r["start"].astype(np.int64) % max(nS, 1)), 0)
    k1 = np.where(tail, cursor, qrank)
    k2 = np.where(tail, qrank, sp)
    order = np.lexsort((k2, k1, end))                        # stable: the order inside one (site, query, cursor) slot is the rank's own
    out = np.zeros(len(r), MERGED_DTYPE)
    for f in ("ai", "bi", "start", "end"):
        out[f] = r[f][order]
    out["sparse"] = sp[order]
    return out


def merge_events(events):
    """no-match events of all ranks -> the reference's log order (site, query rank, dense before sparse), tags stripped"""
    parts = [np.asarray(e, dtype=np.int32).reshape(-1, 4) for e in events if len(e)]
    if not parts:
        return np.zeros((0, 4), np.int32)
    ev = np.concatenate(parts)
    order = np.lexsort((ev[:, 3] & 1, ev[:, 3] >> 1, ev[:, 2]))
    ev = ev[order].copy()
    ev[:, 3] &= 1
    return ev


def _gather_rows(rows, device=None):
    """all ranks' int32 row blocks on every rank (all-gather of padded tensors; sizes first)"""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [rows]
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    width = rows.shape[1]
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = max(int(s.item()) for s in sizes)
    mine = torch.zeros((max(cap, 1), width), dtype=torch.int32, device=dev)
    if rows.shape[0]:
        mine[: rows.shape[0]] = torch.from_numpy(rows).to(dev)
    bufs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return [b[: int(s.item())].cpu().numpy() for b, s in zip(bufs, sizes)]


def match_sweep_sharded(eng, pz, N, qz, Mq, nSparse=0, pStart=None, qStart=None, device=None):
    """the whole job on every rank of the default process group: sweep the own range, exchange, merge.
    Returns (records in the reference's order, no-match events in log order, n_nomatch, (nTot, totLen)) on every rank."""
    import torch.distributed as dist
    from . import dist as pdist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if on else 0
    world = dist.get_world_size() if on else 1
    lo, hi = plan_ranges(Mq, world)[rank]
    recs, ev, nom, tot = run_range(eng, pz, N, qz, Mq, lo, hi, nSparse, pStart, qStart)
    flat = np.stack([recs[f] for f in ("ai", "bi", "start", "end", "sparse")], axis=1) if len(recs) else np.zeros((0, 5), np.int32)
    all_recs = _gather_rows(flat, device)
    all_ev = _gather_rows(np.asarray(ev, np.int32).reshape(-1, 4), device)
    streams = []
    for a in all_recs:
        s = np.zeros(len(a), MERGED_DTYPE)
        for i, f in enumerate(("ai", "bi", "start", "end", "sparse")):
            s[f] = a[:, i]
        streams.append(s)
    merged = merge_streams(streams, N, nSparse)
    nom_all = int(pdist.sum_over_ranks(float(nom), device=device)) if on else nom
    tot_all = (int(pdist.sum_over_ranks(float(tot[0]), device=device)), int(pdist.sum_over_ranks(float(tot[1]), device=device))) if on else tot
    return merged, merge_events(all_ev), nom_all, tot_all
