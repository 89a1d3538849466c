"""numpy model of the tile/summary formulation used by step_kernel (pbwt_amd/csrc/pbwt_kernels.h).

It mirrors the kernel's arithmetic (per-tile summaries cnt0/last0/last1/maxd built by the previous
step, carries from partial-tile reads + whole-tile maxima, local recurrence, destination offsets)
with a tiny tile size so every cross-tile path is exercised on the CPU.  It is a test aid that
documents the algorithm; the product is the HIP kernel.
"""
import numpy as np


def summaries(y, d, M, T):
    W = (M + T - 1) // T
    cnt0 = np.zeros(W, np.int64); last0 = np.zeros(W, np.int64); last1 = np.zeros(W, np.int64); maxd = np.zeros(W, np.int64)
    for i in range(M):
        w = i // T
        if y[i] == 0:
            cnt0[w] += 1; last0[w] = max(last0[w], i + 1)
        else:
            last1[w] = max(last1[w], i + 1)
        maxd[w] = max(maxd[w], d[i])
    return cnt0, last0, last1, maxd


def step_tiles(a, d, y, k, T, summ):
    """one site; returns a', d' computed tile by tile using only tile-local data + summaries"""
    M = len(a)
    W = (M + T - 1) // T
    cnt0, last0, last1, maxd = summ
    C = int(cnt0.sum())
    a2 = np.zeros(M, np.int64); d2 = np.zeros(M + 1, np.int64)
    for w in range(W):
        S = w * T
        Zw = int(cnt0[:w].sum())
        l = [int(last0[:w].max()) if w else 0, int(last1[:w].max()) if w else 0]
        carry = [0, 0]
        for b in (0, 1):
            if l[b] == 0:
                carry[b] = k + 1
            else:
                tl = (l[b] - 1) // T
                m = 0
                for jn in range(tl + 1, w):
                    m = max(m, int(maxd[jn]))
                hi = min((tl + 1) * T, S)
                for p in range(l[b], hi):
                    m = max(m, int(d[p]))
                carry[b] = m
        p, q = carry
        zi, oi = Zw, C + (S - Zw)
        for i in range(S, min(S + T, M)):
            p = max(p, int(d[i])); q = max(q, int(d[i]))
            if y[i] == 0:
                a2[zi] = a[i]; d2[zi] = p; zi += 1; p = 0
            else:
                a2[oi] = a[i]; d2[oi] = q; oi += 1; q = 0
    d2[0] = k + 2; d2[M] = k + 2
    return a2, d2


# ------------------------------------------------------------------------------------------------
# two sites per launch (step2_kernel): key = b0 | b1<<1 (b0 = allele at site k, b1 at site k+1).
# Everything at both levels is a static function of (keys, d_k):
#   same-key predecessor (level-0 order)  -> range max of d_k;  no predecessor -> k+1+msb(key^key')
#   with key' the nearest lower non-empty bucket; level-1 value = min over the two keys sharing b0.
def summaries2(key, d, M, T):
    W = (M + T - 1) // T
    c = np.zeros((W, 4), np.int64); last = np.zeros((W, 4), np.int64); maxd = np.zeros(W, np.int64)
    for i in range(M):
        w = i // T
        c[w, key[i]] += 1
        last[w, key[i]] = max(last[w, key[i]], i + 1)
        maxd[w] = max(maxd[w], d[i])
    return c, last, maxd


def step2_tiles(a, d, key, k, T, summ):
    """sites k and k+1 in one pass; returns (a1, d1) = state before site k+1, (a2, d2) = before k+2"""
    M = len(a)
    W = (M + T - 1) // T
    c, last, maxd = summ
    tot = c.sum(axis=0)
    C1 = int(tot[0] + tot[2])                       # zeros of site k (keys with b0 == 0)
    G2 = np.concatenate([[0], np.cumsum(tot)])[:4]  # bucket bases at level 2, order 00,01,10,11 (b1 major)
    a1 = np.zeros(M, np.int64); d1 = np.zeros(M + 1, np.int64)
    a2 = np.zeros(M, np.int64); d2 = np.zeros(M + 1, np.int64)
    for w in range(W):
        S = w * T
        before = c[:w].sum(axis=0) if w else np.zeros(4, np.int64)
        l = [int(last[:w, q].max()) if w else 0 for q in range(4)]
        carry = [0] * 4
        for q in range(4):
            if l[q]:
                tl = (l[q] - 1) // T
                m = 0
                for jn in range(tl + 1, w):
                    m = max(m, int(maxd[jn]))
                for p in range(l[q], min((tl + 1) * T, S)):
                    m = max(m, int(d[p]))
                carry[q] = m
        # running state inside the tile: t[q] = max d since the last key-q element (INF when none yet here)
        seen = [False] * 4
        t = [0] * 4
        allm = 0
        cnt = [0] * 4
        for i in range(S, min(S + T, M)):
            q = int(key[i]); di = int(d[i])
            b0 = q & 1
            # effective "max d since last key-q' element" for every key, including earlier tiles
            def eff(qq):
                if seen[qq]:
                    return t[qq], True
                if l[qq]:
                    return max(carry[qq], allm), True
                return None, False
            # level 2
            v, ex = eff(q)
            if ex:
                dd2 = max(v, di)
            else:
                lower = [qq for qq in range(q) if tot[qq] > 0]
                dd2 = (k + 1 + ((q ^ lower[-1]).bit_length() - 1)) if lower else 0   # pos 0 gets the sentinel
            # level 1: the later of the two keys sharing b0 = the smaller of the two maxima
            cand = [eff(qq) for qq in (b0, b0 + 2)]
            vals = [vv for vv, e2 in cand if e2]
            dd1 = max(min(vals), di) if vals else k + 1
            zr = cnt[0] + cnt[2]; orr = cnt[1] + cnt[3]
            Zw1 = int(before[0] + before[2])
            pos1 = (C1 + (S - Zw1) + orr) if b0 else (Zw1 + zr)
            pos2 = int(G2[q] + before[q] + cnt[q])
            a1[pos1] = a[i]; d1[pos1] = dd1
            a2[pos2] = a[i]; d2[pos2] = dd2
            # update running state
            allm = max(allm, di)
            for qq in range(4):
                if qq == q:
                    t[qq] = 0; seen[qq] = True
                elif seen[qq]:
                    t[qq] = max(t[qq], di)
            cnt[q] += 1
    d1[0] = k + 2; d1[M] = k + 2
    d2[0] = k + 3; d2[M] = k + 3
    return (a1, d1), (a2, d2)


# ------------------------------------------------------------------------------------------------
# B sites per step, level-B output only ("skeleton" step; stepB kernels K1/K2/K3):
#   key = alleles at sites k..k+B-1 (bit j = site k+j); a_{k+B} = stable sort of a_k by key;
#   same-key predecessor (level-0 order) -> range max of d_k; none -> k+1+msb(key ^ lower non-empty key).
def stepB_tiles(a, d, key, k, B, T):
    M = len(a)
    W = (M + T - 1) // T
    K = 1 << B
    # K1: per tile, per key: count and the max of d_k after the key's last occurrence (whole-tile max if absent)
    cnt = np.zeros((W, K), np.int64); tail = np.zeros((W, K), np.int64)
    for w in range(W):
        lo, hi = w * T, min((w + 1) * T, M)
        kk = key[lo:hi]; dd = d[lo:hi]
        for q in range(K):
            idx = np.nonzero(kk == q)[0]
            cnt[w, q] = len(idx)
            tail[w, q] = (dd[idx[-1] + 1:].max() if len(idx) and idx[-1] + 1 < len(dd) else 0) if len(idx) else dd.max()
    # K2: per key, scan over tiles
    before = np.zeros((W, K), np.int64); carry = -np.ones((W, K), np.int64)
    for q in range(K):
        run, ex, c = 0, False, 0
        for w in range(W):
            before[w, q] = run
            carry[w, q] = c if ex else -1
            if cnt[w, q]:
                ex = True; c = tail[w, q]
            elif ex:
                c = max(c, tail[w, q])
            run += cnt[w, q]
    total = cnt.sum(axis=0)
    G = np.concatenate([[0], np.cumsum(total)])[:K]
    lower = -np.ones(K, np.int64)
    last = -1
    for q in range(K):
        lower[q] = last
        if total[q]:
            last = q
    # K3: per tile
    a2 = np.zeros(M, np.int64); d2 = np.zeros(M + 1, np.int64)
    for w in range(W):
        lo, hi = w * T, min((w + 1) * T, M)
        seen_at = {}
        rank = {}
        for i in range(lo, hi):
            q = int(key[i])
            r = rank.get(q, 0)
            if q in seen_at:
                dd = int(d[seen_at[q] + 1: i + 1].max())
            elif carry[w, q] >= 0:
                dd = max(int(carry[w, q]), int(d[lo: i + 1].max()))
            elif lower[q] >= 0:
                dd = k + 1 + ((q ^ int(lower[q])).bit_length() - 1)
            else:
                dd = 0
            pos = int(G[q] + before[w, q] + r)
            a2[pos] = a[i]; d2[pos] = dd
            seen_at[q] = i; rank[q] = r + 1
    d2[0] = k + B + 1; d2[M] = k + B + 1
    return a2, d2


# ------------------------------------------------------------------------------------------------
# the fill (skel_fill_kernel): the states between two skeleton states out of the SAME per-tile tables.
# State k+j (1 <= j < B) is the stable sort of state k by the low j bits of the keys; for a j-bit key q
#   count  = sum of the counts of the B-bit keys with low bits q            (per tile)
#   before = sum of their `before`;  G from the folded totals
#   carry  = min over those keys' carries that exist (a later last occurrence has the smaller suffix max)
#   lower  = nearest lower non-empty j-bit key
def fillB_tiles(a, d, key, k, B, T):
    """returns [(a_j, d_j) for j = 1..B-1], each computed tile by tile from level-B tables folded down"""
    M = len(a)
    W = (M + T - 1) // T
    K = 1 << B
    cnt = np.zeros((W, K), np.int64); tail = np.zeros((W, K), np.int64)
    for w in range(W):
        lo, hi = w * T, min((w + 1) * T, M)
        kk = key[lo:hi]; dd = d[lo:hi]
        for q in range(K):
            idx = np.nonzero(kk == q)[0]
            cnt[w, q] = len(idx)
            tail[w, q] = (dd[idx[-1] + 1:].max() if len(idx) and idx[-1] + 1 < len(dd) else 0) if len(idx) else dd.max()
    before = np.zeros((W, K), np.int64); carry = -np.ones((W, K), np.int64)
    for q in range(K):
        run, ex, c = 0, False, 0
        for w in range(W):
            before[w, q] = run
            carry[w, q] = c if ex else -1
            if cnt[w, q]:
                ex = True; c = tail[w, q]
            elif ex:
                c = max(c, tail[w, q])
            run += cnt[w, q]
    total = cnt.sum(axis=0)
    out = []
    for j in range(1, B):
        Kj = 1 << j
        fold = lambda arr: arr.reshape(-1, K >> j, Kj).sum(axis=1) if arr.ndim == 2 else arr.reshape(K >> j, Kj).sum(axis=0)
        before_j = fold(before); total_j = fold(total)
        cj = carry.reshape(W, K >> j, Kj).astype(np.float64)
        cj[cj < 0] = np.inf
        carry_j = cj.min(axis=1); carry_j[np.isinf(carry_j)] = -1
        G = np.concatenate([[0], np.cumsum(total_j)])[:Kj]
        lower = -np.ones(Kj, np.int64); last = -1
        for q in range(Kj):
            lower[q] = last
            if total_j[q]:
                last = q
        a2 = np.zeros(M, np.int64); d2 = np.zeros(M + 1, np.int64)
        for w in range(W):
            lo, hi = w * T, min((w + 1) * T, M)
            seen_at, rank = {}, {}
            for i in range(lo, hi):
                q = int(key[i]) & (Kj - 1)
                r = rank.get(q, 0)
                if q in seen_at:
                    dd = int(d[seen_at[q] + 1: i + 1].max())
                elif carry_j[w, q] >= 0:
                    dd = max(int(carry_j[w, q]), int(d[lo: i + 1].max()))
                elif lower[q] >= 0:
                    dd = k + 1 + ((q ^ int(lower[q])).bit_length() - 1)
                else:
                    dd = 0
                pos = int(G[q] + before_j[w, q] + r)
                a2[pos] = a[i]; d2[pos] = dd
                seen_at[q] = i; rank[q] = r + 1
        d2[0] = k + j + 1; d2[M] = k + j + 1
        out.append((a2, d2))
    return out


# ------------------------------------------------------------------------------------------------
# read side (skel_keys_sorted_kernel): the B-bit key of POSITION i of state k from the columns in PBWT order,
# through the LF-mapping: bit j = y_{k+j}[p_j], p_0 = i, p_{j+1} = y ? c + p_j - u(p_j) : u(p_j)
def keys_from_sorted_columns(ys, B):
    """ys[j] = y_{k+j} in the order of a_{k+j} (what unpack3 yields); returns the key of every position of state k"""
    M = len(ys[0])
    pos = np.arange(M)
    key = np.zeros(M, np.int64)
    for j in range(B):
        y = ys[j].astype(np.int64)
        u = np.concatenate([[0], np.cumsum(1 - y)])          # zeros before each position
        c = int(u[M])
        bit = y[pos]
        key |= bit << j
        pos = np.where(bit == 1, c + pos - u[pos], u[pos])
    return key


# ------------------------------------------------------------------------------------------------
# the fill as a SEQUENTIAL tile-local recurrence (skel_fillseq_kernel): the tile is carried through the B-1 sub-steps in its own
# sorted order.  At level j the tile's elements stand sorted by their j-bit keys; the elements of one key ("run") are contiguous
# in the global state k+j too, so sub-step j -> j+1 is ONE stable partition of the local array by bit j plus a segmented max:
#   stretch = maximal group of neighbours with equal (j+1)-bit key (inside a run the stretches alternate between the two bit values);
#   inside a stretch d' = d (the same-bit predecessor is the neighbour); the head of a stretch takes max(d, max d of the stretch
#   before it) when that stretch is not the first of its run, else — no same-bit predecessor in the tile's run — the folded
#   skeleton tables decide: carry of the (j+1)-bit key (max with the local maximum since the run's start) or the key-difference value.
def fillS_tiles(d, key, k, B, T):
    """returns [(dest_j, d_j) for j = 1..B-1]: d_j = the divergences of state k+j (global order, sentinels at 0 and M) and
    dest_j[i] = position in state k+j of the element at position i of state k — computed tile by tile, sub-step by sub-step"""
    M = len(key)
    W = (M + T - 1) // T
    K = 1 << B
    cnt = np.zeros((W, K), np.int64); tail = np.zeros((W, K), np.int64)
    for w in range(W):
        lo, hi = w * T, min((w + 1) * T, M)
        kk = key[lo:hi]; dd = d[lo:hi]
        for q in range(K):
            idx = np.nonzero(kk == q)[0]
            cnt[w, q] = len(idx)
            tail[w, q] = (dd[idx[-1] + 1:].max() if len(idx) and idx[-1] + 1 < len(dd) else 0) if len(idx) else dd.max()
    before = np.zeros((W, K), np.int64); carry = -np.ones((W, K), np.int64)
    for q in range(K):
        run, ex, c = 0, False, 0
        for w in range(W):
            before[w, q] = run
            carry[w, q] = c if ex else -1
            if cnt[w, q]:
                ex = True; c = tail[w, q]
            elif ex:
                c = max(c, tail[w, q])
            run += cnt[w, q]
    total = cnt.sum(axis=0)
    # per level (tile-independent): bucket bases G and the value of an element whose key has no earlier occurrence at all
    GB, base, before_l, carry_l = {}, {}, {}, {}
    for j in range(1, B):
        Kj = 1 << j
        total_j = total.reshape(K >> j, Kj).sum(axis=0)
        GB[j] = np.concatenate([[0], np.cumsum(total_j)])[:Kj]
        bs = np.zeros(Kj, np.int64); last = -1
        for q in range(Kj):
            bs[q] = (k + 1 + ((q ^ last).bit_length() - 1)) if last >= 0 else 0
            if total_j[q]:
                last = q
        base[j] = bs
        before_l[j] = before.reshape(W, K >> j, Kj).sum(axis=1)
        cj = carry.reshape(W, K >> j, Kj).astype(np.float64); cj[cj < 0] = np.inf
        cm = cj.min(axis=1); cm[np.isinf(cm)] = -1
        carry_l[j] = cm.astype(np.int64)
    outs = [(np.zeros(M, np.int64), np.zeros(M + 1, np.int64)) for _ in range(1, B)]
    for w in range(W):
        lo, hi = w * T, min((w + 1) * T, M)
        n = hi - lo
        dc = d[lo:hi].astype(np.int64).copy()
        kc = key[lo:hi].astype(np.int64).copy()
        src = np.arange(lo, hi)                                   # model only: which state-k position stands here
        for j in range(0, B - 1):                                 # sub-step j -> j + 1
            mj, mj1 = (1 << j) - 1, (1 << (j + 1)) - 1
            v = (kc >> j) & 1
            P1 = np.concatenate([[0], np.cumsum(v)])[:n]          # ones before each position
            Z = n - int(v.sum())
            ni = np.where(v == 1, Z + P1, np.arange(n) - P1)
            Rst = np.ones(n, bool); Rst[1:] = ((kc[1:] ^ kc[:-1]) & mj) != 0        # run start (level j)
            H = np.ones(n, bool); H[1:] = ((kc[1:] ^ kc[:-1]) & mj1) != 0           # stretch head
            S = np.zeros(n, np.int64); Rh = np.zeros(n, bool)     # segmented max over the stretch; "this stretch starts its run"
            for i in range(n):
                if H[i]:
                    S[i] = dc[i]; Rh[i] = Rst[i]
                else:
                    S[i] = max(S[i - 1], dc[i]); Rh[i] = Rh[i - 1]
            nd = dc.copy()
            off = {}
            for i in range(n):
                if not H[i]:
                    continue
                kq = int(kc[i] & mj1)
                if not Rst[i] and not Rh[i - 1]:
                    nd[i] = max(dc[i], S[i - 1])
                else:                                             # first of its (j+1)-bit key in the tile
                    loc = dc[i] if Rst[i] else max(dc[i], S[i - 1])
                    c = carry_l[j + 1][w, kq]
                    nd[i] = max(c, loc) if c >= 0 else base[j + 1][kq]
                    off[kq] = GB[j + 1][kq] + before_l[j + 1][w, kq] - ni[i]
            dn = np.zeros(n, np.int64); kn = np.zeros(n, np.int64); sn = np.zeros(n, np.int64)
            dn[ni] = nd; kn[ni] = kc; sn[ni] = src
            dc, kc, src = dn, kn, sn
            dest, dj = outs[j]
            for x in range(n):
                pos = x + off[int(kc[x] & mj1)]
                dj[pos] = dc[x]; dest[src[x]] = pos
        # nothing else leaves the tile
    for j in range(1, B):
        outs[j - 1][1][0] = k + j + 1; outs[j - 1][1][M] = k + j + 1
    return outs
