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
