"""CPU: the tile/summary formulation the HIP step kernel implements (tests/tile_model.py) gives the
reference's a[]/d[] at every site, with tiny tiles so all cross-tile carries are exercised."""
import numpy as np
import pytest

from tile_model import step_tiles, summaries


@pytest.mark.parametrize("M,N,kind,T", [(37, 60, 1, 8), (100, 120, 0, 8), (64, 50, 1, 16), (257, 150, 0, 32), (1000, 60, 0, 64)])
def test_tile_model_matches_oracle(orc, M, N, kind, T):
    bits = orc.synth_bitcols(M, N, seed=M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a = np.arange(M)
    d = np.zeros(M + 1, np.int64)
    d[0] = d[M] = 1
    for k in range(N):
        assert np.array_equal(a, o["a_dump"][k]) and np.array_equal(d, o["d_dump"][k])
        y = hap[k][a]
        a, d = step_tiles(a, d, y, k, T, summaries(y, d, M, T))
    assert np.array_equal(a, o["a_dump"][N]) and np.array_equal(d, o["d_dump"][N])


@pytest.mark.parametrize("M,N,kind,T", [(37, 60, 1, 8), (100, 121, 0, 8), (257, 90, 0, 32), (500, 41, 1, 64)])
def test_two_site_tile_model_matches_oracle(orc, M, N, kind, T):
    """two sites per launch (step2_kernel's formulation): both output levels against the oracle"""
    from tile_model import step2_tiles, summaries2
    bits = orc.synth_bitcols(M, N, seed=3 * M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a, d = o["a_dump"][0].astype(np.int64), o["d_dump"][0].astype(np.int64)
    for k in range(0, N - 1, 2):
        key = hap[k][a] | (hap[k + 1][a] << 1)
        (a1, d1), (a, d) = step2_tiles(a, d, key, k, T, summaries2(key, d, M, T))
        assert np.array_equal(a1, o["a_dump"][k + 1]) and np.array_equal(d1, o["d_dump"][k + 1])
        assert np.array_equal(a, o["a_dump"][k + 2]) and np.array_equal(d, o["d_dump"][k + 2])


@pytest.mark.parametrize("M,N,kind,T,B", [(37, 64, 1, 8, 8), (100, 120, 0, 8, 8), (257, 96, 0, 32, 8), (300, 60, 0, 16, 3),
                                         (1000, 40, 0, 64, 8), (64, 35, 1, 16, 5)])
def test_skeleton_tile_model_matches_oracle(orc, M, N, kind, T, B):
    """B sites per step (the skeleton chain K1/K2/K3: per-tile key histograms and tails, per-key scan
    over tiles, ranks + range maxima): the state every B sites against the oracle"""
    from tile_model import stepB_tiles
    bits = orc.synth_bitcols(M, N, seed=5 * M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M).astype(np.int64)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a, d = o["a_dump"][0].astype(np.int64), o["d_dump"][0].astype(np.int64)
    for k in range(0, N - B + 1, B):
        key = np.zeros(M, np.int64)
        for j in range(B):
            key |= hap[k + j][a] << j
        a, d = stepB_tiles(a, d, key, k, B, T)
        assert np.array_equal(a, o["a_dump"][k + B]), "a at site %d" % (k + B)
        assert np.array_equal(d, o["d_dump"][k + B]), "d at site %d" % (k + B)


@pytest.mark.parametrize("M,N,kind,T,B", [(37, 64, 1, 8, 8), (100, 72, 0, 8, 8), (257, 48, 0, 32, 8), (300, 30, 0, 16, 3), (64, 35, 1, 16, 5)])
def test_fill_model_matches_oracle(orc, M, N, kind, T, B):
    """the fill (skel_fill_kernel's formulation): every state between two skeleton states from the skeleton's per-tile
    tables folded down bit by bit (counts add, carries min, lower key per level)"""
    from tile_model import fillB_tiles, stepB_tiles
    bits = orc.synth_bitcols(M, N, seed=9 * M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M).astype(np.int64)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a, d = o["a_dump"][0].astype(np.int64), o["d_dump"][0].astype(np.int64)
    for k in range(0, N - B + 1, B):
        key = np.zeros(M, np.int64)
        for j in range(B):
            key |= hap[k + j][a] << j
        for j, (aj, dj) in enumerate(fillB_tiles(a, d, key, k, B, T), start=1):
            assert np.array_equal(aj, o["a_dump"][k + j]), "a at site %d" % (k + j)
            assert np.array_equal(dj, o["d_dump"][k + j]), "d at site %d" % (k + j)
        a, d = stepB_tiles(a, d, key, k, B, T)


@pytest.mark.parametrize("M,N,kind,B", [(37, 64, 1, 8), (300, 40, 0, 8), (1000, 24, 0, 8), (64, 35, 1, 5)])
def test_read_side_keys_through_lf_mapping(orc, M, N, kind, B):
    """read side: the key of a position follows the LF-mapping through the sorted columns and equals the key the build
    side gathers by haplotype"""
    from tile_model import keys_from_sorted_columns
    bits = orc.synth_bitcols(M, N, seed=11 * M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M).astype(np.int64)
    o = orc.build_bitcols(bits, M, with_d=False, dump_sites=range(N + 1))
    for k in range(0, N - B + 1, B):
        ys = [hap[k + j][o["a_dump"][k + j]] for j in range(B)]          # y_{k+j} in the order of a_{k+j}
        want = np.zeros(M, np.int64)
        for j in range(B):
            want |= hap[k + j][o["a_dump"][k]] << j
        assert np.array_equal(keys_from_sorted_columns(ys, B), want)


@pytest.mark.parametrize("M,N,kind,T,B", [(37, 64, 1, 8, 8), (100, 72, 0, 8, 8), (257, 48, 0, 32, 8), (300, 30, 0, 16, 3), (64, 35, 1, 16, 5),
                                         (700, 32, 0, 64, 8), (1000, 24, 1, 128, 8)])
def test_sequential_fill_model_matches_oracle(orc, M, N, kind, T, B):
    """the fill as a tile-local sequential recurrence (skel_fillseq_kernel's formulation): one stable partition + one segmented max per
    sub-step in the tile's own order, the folded skeleton tables only for the first element of a key in the tile"""
    from tile_model import fillS_tiles, stepB_tiles
    bits = orc.synth_bitcols(M, N, seed=13 * M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M).astype(np.int64)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a, d = o["a_dump"][0].astype(np.int64), o["d_dump"][0].astype(np.int64)
    for k in range(0, N - B + 1, B):
        key = np.zeros(M, np.int64)
        for j in range(B):
            key |= hap[k + j][a] << j
        for j, (dest, dj) in enumerate(fillS_tiles(d, key, k, B, T), start=1):
            aj = np.zeros(M, np.int64); aj[dest] = a
            assert np.array_equal(aj, o["a_dump"][k + j]), "a at site %d" % (k + j)
            assert np.array_equal(dj, o["d_dump"][k + j]), "d at site %d" % (k + j)
        a, d = stepB_tiles(a, d, key, k, B, T)


def test_local_scan_fold_equals_global_scan():
    """skel_k2_local_kernel + sk_fold_aggx (pbwt_k_chain.h) restated in numpy: the per-key scan over the tiles' (count, tail) rows done in
    workgroups of TPW rows — raw prefixes local to the workgroup, one exclusive aggregate row per workgroup — and folded by the reader gives the
    rows skel_k2_wide_kernel writes: (keys before the tile, carry = max d since the key's last earlier occurrence, -1 without one)."""
    rng = np.random.default_rng(5)
    for W, K, TPW in [(70, 16, 32), (33, 8, 32), (129, 4, 64), (5, 32, 32)]:
        cnt = (rng.random((W, K)) < 0.4) * rng.integers(1, 5, (W, K))
        tail = rng.integers(0, 1000, (W, K))                # (a row without the key: its tail is the maximum of the whole tile)
        # global scan (skel_k2_wide_kernel's second pass)
        want = np.zeros((W, K, 2), np.int64)
        for q in range(K):
            ec, et = 0, 0
            for w in range(W):
                want[w, q] = (ec, et if ec else -1)
                et = tail[w, q] if cnt[w, q] else max(et, tail[w, q]); ec += cnt[w, q]
        # local form
        nwg = (W + TPW - 1) // TPW
        loc = np.zeros((W, K, 2), np.int64); agg = np.zeros((nwg, K, 2), np.int64)
        for j in range(nwg):
            for q in range(K):
                lc, lt = 0, 0
                for w in range(j * TPW, min(W, (j + 1) * TPW)):
                    loc[w, q] = (lc, lt)
                    lt = tail[w, q] if cnt[w, q] else max(lt, tail[w, q]); lc += cnt[w, q]
                agg[j, q] = (lc, lt)
        aggx = np.zeros((nwg, K, 2), np.int64)              # the last arriver's fold
        for q in range(K):
            ec, et = 0, 0
            for j in range(nwg):
                aggx[j, q] = (ec, et)
                et = agg[j, q, 1] if agg[j, q, 0] else max(et, agg[j, q, 1]); ec += agg[j, q, 0]
        got = np.zeros_like(want)
        for w in range(W):
            for q in range(K):
                L, R = aggx[w // TPW, q], loc[w, q]          # sk_fold_aggx
                got[w, q] = (L[0] + R[0], R[1] if R[0] else (max(L[1], R[1]) if L[0] else -1))
        assert np.array_equal(got, want), (W, K, TPW)
