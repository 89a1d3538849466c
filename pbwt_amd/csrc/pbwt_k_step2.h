// pbwt_k_step2.h — fallback chains: step_body (one site, E elements per thread) and step2_kernel (two sites per launch, 2-bit keys, tile summaries).
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// per-batch control block in device memory, written by set_ctl_kernel before each batch so that one
// captured graph serves every batch; read with a single scalar load at kernel start.
struct alignas(32) Ctl {
    int kbase;                 // site index of step 0 of this batch
    int n_total;               // sites in the panel (has_next = k+1 < n_total)
    int pad0, pad1;
    const uint32_t *cols;      // bit columns of this batch: column j = site kbase+j
    const uint32_t *zerocol;   // an all-zero column standing in for sites >= n_total (two-site steps)
};

struct StepArgs {
    const int *a_in;  const int *d_in;     // slot j
    int *a_out;       int *d_out;          // slot j+1
    const Ctl *ctl;
    int4 *summ;                            // [3][wpad] {cnt0,last0+1,last1+1,maxd}; step j reads buffer j%3, accumulates (j+1)%3, clears (j+2)%3
    long long *prof;                       // optional phase timestamps [W][8] (NULL = off)
    int wpc;                               // 32-bit words per column
    int j;                                 // step index inside the batch
    int M, W, wpad;
};

#define PBWT_STAMP(idx) do { if (g.prof && t == 0) g.prof[(size_t)w * 8 + (idx)] = (long long)wall_clock64(); } while (0)

// The step kernel is latency-bound, not bandwidth-bound, for M up to ~1M: one wave
// per SIMD executes its instruction stream exactly once, so the launch time is (instructions on
// the longest path) x (~5 cycles) + the dependent memory round trips.  Hence: one or two positions
// per thread, no validity predication on full tiles (FULL), DPP scans, LDS-only barriers, every
// load whose address is known at entry issued first.
template <int E, bool WITH_D, bool SORTED, bool FULL>
__device__ __forceinline__ void step_body(const StepArgs &g, int *s_a, int *s_d, Tup *s_tup, int (*s_red)[6], int (*s_acc)[4]) {
    constexpr int T = BLOCK * E;
    const int j = g.j;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T;                                   // first position of the tile
    PBWT_STAMP(0);

    const int4 *sm_in = g.summ + (size_t)(j % 3) * g.wpad;
    int4 *sm_out = g.summ + (size_t)((j + 1) % 3) * g.wpad;
    int4 *sm_zero = g.summ + (size_t)((j + 2) % 3) * g.wpad;

    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int av[E], dv[E];
    const int base = S + t * E;                            // blocked: thread t owns E consecutive positions
    if constexpr (E % 4 == 0) {
        const int4 *pa = reinterpret_cast<const int4 *>(g.a_in + base);   // arrays are padded to W*T
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const int4 v = pa[q];
            av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        if (WITH_D) {
            const int4 *pd = reinterpret_cast<const int4 *>(g.d_in + base);
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const int4 v = pd[q];
                dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { av[e] = g.a_in[base + e]; if (WITH_D) dv[e] = g.d_in[base + e]; }
    }
    constexpr int SPT = 4;                                 // summaries per thread (W <= 1024)
    int r_cnt[SPT], r_l0[SPT], r_l1[SPT], r_md[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        r_cnt[q] = 0; r_l0[q] = 0; r_l1[q] = 0; r_md[q] = 0;
        if (jn < W) { const int4 sv = sm_in[jn]; r_cnt[q] = sv.x; r_l0[q] = sv.y; r_l1[q] = sv.z; r_md[q] = sv.w; }
    }
    if (t < 16) s_acc[t >> 2][t & 3] = 0;

    const int k = ctl.kbase + j;
    const bool has_next = (k + 1 < ctl.n_total);           // the panel has a site k+1
    const uint32_t *col_next = ctl.cols + (size_t)(j + 1) * g.wpc;

    // ---- own alleles (tags) and, in gather mode, the next-site allele of each haplotype ----
    unsigned ybits = 0, vbits = FULL ? ((1u << E) - 1u) : 0u, nbits = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned y = ((unsigned)av[e]) >> 31;
        av[e] &= AMASK;
        if (FULL) ybits |= y << e;
        else if (base + e < M) { vbits |= 1u << e; ybits |= y << e; }
    }
    if (!SORTED && has_next) {
        unsigned wd[E];
#pragma unroll
        for (int e = 0; e < E; ++e) wd[e] = (FULL || ((vbits >> e) & 1u)) ? col_next[(unsigned)av[e] >> 5] : 0u;
#pragma unroll
        for (int e = 0; e < E; ++e) nbits |= ((wd[e] >> (av[e] & 31)) & 1u) << e;
    }

    // ---- B: tile summaries of this site -> zero offset, total zeros, last-allele positions ----
    int sumBefore = 0, total = 0, l0 = 0, l1 = 0;
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        total += r_cnt[q];
        if (jn < w) { sumBefore += r_cnt[q]; l0 = max(l0, r_l0[q]); l1 = max(l1, r_l1[q]); }
    }
    sumBefore = wave_sum(sumBefore); total = wave_sum(total);
    if (WITH_D) { l0 = wave_max(l0); l1 = wave_max(l1); }
    if (lane == 0) { s_red[wv][0] = sumBefore; s_red[wv][1] = total; s_red[wv][2] = l0; s_red[wv][3] = l1; }
    lds_barrier();
    sumBefore = 0; total = 0; l0 = 0; l1 = 0;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) {
        sumBefore += s_red[q][0]; total += s_red[q][1];
        l0 = max(l0, s_red[q][2]); l1 = max(l1, s_red[q][3]);
    }
    const int Zw = sumBefore;                              // zeros before this tile
    const int C = total;                                   // zeros in the whole column (u->c)
    PBWT_STAMP(1);

    // carry_b = max d over [l_b, S): the positions after the last allele-b element before the tile
    // = direct reads in the tile holding position l_b - 1, plus whole-tile maxima in between.
    int cw, nvalid;                                        // zeros / valid positions in this tile
    if (WITH_D) {
        int m0 = 0, m1 = 0;
        const int tl0 = l0 ? (l0 - 1) / T : -1, tl1 = l1 ? (l1 - 1) / T : -1;
        int pd0[E], pd1[E];
        const int hi0 = l0 ? min((tl0 + 1) * T, S) : 0, hi1 = l1 ? min((tl1 + 1) * T, S) : 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {                      // issue the dependent loads first
            const int p0 = l0 + t + e * BLOCK, p1 = l1 + t + e * BLOCK;
            pd0[e] = (l0 && p0 < hi0) ? g.d_in[p0] : 0;
            pd1[e] = (l1 && p1 < hi1) ? g.d_in[p1] : 0;
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int jn = t + q * BLOCK;
            if (jn < w) {
                if (l0 && jn > tl0) m0 = max(m0, r_md[q]);
                if (l1 && jn > tl1) m1 = max(m1, r_md[q]);
            }
        }
        // ---- C (overlaps the loads above): thread-local carry tuple ----
        Tup me = Tup{0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                const int d = dv[e];
                me.all = max(me.all, d);
                if (!((ybits >> e) & 1u)) { me.t0 = 0; me.t1 = max(me.t1, d); ++me.c0; }
                else                      { me.t1 = 0; me.t0 = max(me.t0, d); ++me.c1; }
            }
        }
        Tup tot;
        const Tup pre = block_scan_tup<true>(me, s_tup, tot);
        PBWT_STAMP(2);
#pragma unroll
        for (int e = 0; e < E; ++e) { m0 = max(m0, pd0[e]); m1 = max(m1, pd1[e]); }
        m0 = wave_max(m0); m1 = wave_max(m1);
        if (lane == 0) { s_red[wv][4] = m0; s_red[wv][5] = m1; }
        lds_barrier();
        m0 = 0; m1 = 0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { m0 = max(m0, s_red[q][4]); m1 = max(m1, s_red[q][5]); }
        const int carry0 = l0 ? m0 : k + 1;                // nothing before: p starts at k+1 (pbwtCore.c:489)
        const int carry1 = l1 ? m1 : k + 1;
        PBWT_STAMP(3);
        cw = tot.c0; nvalid = tot.c0 + tot.c1;
        int p = pre.c0 ? pre.t0 : max(carry0, pre.all);
        int q1 = pre.c1 ? pre.t1 : max(carry1, pre.all);
        int zi = pre.c0, oi = cw + pre.c1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                int ldst, dn;
                if (!((ybits >> e) & 1u)) { dn = max(p, dv[e]); p = 0; q1 = max(q1, dv[e]); ldst = zi++; }
                else                      { dn = max(q1, dv[e]); q1 = 0; p = max(p, dv[e]); ldst = oi++; }
                s_a[ldst] = av[e] | (int)(((nbits >> e) & 1u) << 31);
                s_d[ldst] = dn;
            }
        }
    } else {
        Tup me = Tup{0, 0, 0, 0, 0};
        me.c0 = __popc(vbits & ~ybits); me.c1 = __popc(vbits & ybits);
        Tup tot;
        const Tup pre = block_scan_tup<false>(me, s_tup, tot);
        PBWT_STAMP(2);
        PBWT_STAMP(3);
        cw = tot.c0; nvalid = tot.c0 + tot.c1;
        int zi = pre.c0, oi = cw + pre.c1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (FULL || (vbits & (1u << e))) {
                const int ldst = ((ybits >> e) & 1u) ? oi++ : zi++;
                s_a[ldst] = av[e] | (int)(((nbits >> e) & 1u) << 31);
            }
        }
    }
    lds_barrier();
    PBWT_STAMP(4);

    // ---- D: coalesced write-out in destination order + summaries of site k+1 ----
    const int onesBefore = S - Zw;                         // every earlier tile is full
    const int oneBase = C + onesBefore;                    // destination of this tile's first one
    const int tz = Zw / T, to = oneBase / T;               // first destination tile of each stream
    int mdl[4] = {0, 0, 0, 0};                             // per-lane max d' per destination slot (E > 2)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int l = e * BLOCK + t;
        const bool valid = FULL || (l < nvalid);
        int P = 0, slot = -1, dn = 0;
        unsigned tag = 0;
        if (valid) {
            int a = s_a[l];
            const bool one = l >= cw;
            P = one ? oneBase + (l - cw) : Zw + l;
            slot = one ? 2 + (int)((unsigned)P / T - to) : (int)((unsigned)P / T - tz);
            if (SORTED) {
                if (has_next) tag = (col_next[(unsigned)P >> 5] >> (P & 31)) & 1u;
                a |= (int)(tag << 31);
            } else tag = (unsigned)a >> 31;
            g.a_out[P] = a;
            if (WITH_D) {
                dn = s_d[l];
                if (P == 0) dn = k + 2;                    // sentinel (pbwtCore.c:507)
                g.d_out[P] = dn;
            }
        }
        if (has_next) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned long long mk = __ballot(slot == s);
                if (mk) {                                  // wave-uniform
                    const unsigned long long ones = __ballot(slot == s && tag);
                    const unsigned long long zeros = mk & ~ones;
                    int md = 0;
                    if (WITH_D) {
                        if (E > 2) mdl[s] = max(mdl[s], (slot == s) ? dn : 0);
                        else md = wave_max((slot == s) ? dn : 0);
                    }
                    // lanes of one slot are consecutive positions: P(lane) = P(first) + lane - first
                    const int first = __ffsll((long long)mk) - 1;
                    const int Pf = __builtin_amdgcn_readlane(P, first);
                    if (lane == 0) {
                        atomicAdd(&s_acc[s][0], __popcll(zeros));
                        if (WITH_D) {
                            if (zeros) atomicMax(&s_acc[s][1], Pf + (63 - __clzll(zeros)) - first + 1);
                            if (ones) atomicMax(&s_acc[s][2], Pf + (63 - __clzll(ones)) - first + 1);
                            if (E <= 2) atomicMax(&s_acc[s][3], md);
                        }
                    }
                }
            }
        }
    }
    if (WITH_D && has_next && E > 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = wave_max(mdl[s]);
            if (lane == 0 && m) atomicMax(&s_acc[s][3], m);
        }
    }
    if (WITH_D && w == W - 1 && t == 0) g.d_out[M] = k + 2;
    lds_barrier();
    PBWT_STAMP(5);
    if (has_next && t < 4) {
        const int s = t;
        const int dt = (s < 2 ? tz : to) + (s & 1);
        if (dt < W) {
            int *so = reinterpret_cast<int *>(sm_out + dt);
            const int c0 = s_acc[s][0];
            if (c0) atomicAdd(so, c0);
            if (WITH_D) {
                if (s_acc[s][1]) atomicMax(so + 1, s_acc[s][1]);
                if (s_acc[s][2]) atomicMax(so + 2, s_acc[s][2]);
                if (s_acc[s][3]) atomicMax(so + 3, s_acc[s][3]);
            }
        }
    }
    if (t == 0) sm_zero[w] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(6);
}

// ---------------------------------------------------------------------------------------------
// step2_kernel: TWO sites per launch (gather mode, E = 1).  a_{k+2} is the stable 4-way partition of
// a_k by the key q = b0 | b1<<1 (alleles at sites k, k+1), and every divergence at both levels is a
// static function of (keys, d_k) (tests/tile_model.py::step2_tiles):
//   predecessor with the same key in level-0 order  -> range max of d_k over (pred, e]
//   no such predecessor                             -> k + 1 + msb(q ^ q'), q' = nearest lower non-empty key
//   level-1 value (same b0)                         -> the smaller of the two keys' running maxima
// so the per-launch fixed cost (launch gap + first round trip) is paid once per two sites.
// Tile summaries for the next PAIR of sites, per tile: c[4] (keys), last[4] (+1), maxd — 3 int4,
// all commutative, accumulated by the launch that scatters into that order.
struct Tup4 { int c[4]; int t[4]; int all; };

__device__ __forceinline__ Tup4 tup4_id() { Tup4 r; for (int q = 0; q < 4; ++q) { r.c[q] = 0; r.t[q] = 0; } r.all = 0; return r; }
__device__ __forceinline__ Tup4 tup4_combine(const Tup4 &L, const Tup4 &R) {
    Tup4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) { o.c[q] = L.c[q] + R.c[q]; o.t[q] = R.c[q] ? R.t[q] : max(L.t[q], R.all); }
    o.all = max(L.all, R.all);
    return o;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ Tup4 tup4_dpp(const Tup4 &v) {
    Tup4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.c[q] = dpp_mov<CTRL, ROWMASK>(0, v.c[q]); r.t[q] = dpp_mov<CTRL, ROWMASK>(0, v.t[q]); }
    r.all = dpp_mov<CTRL, ROWMASK>(0, v.all);
    return r;
}
// wave totals -> every wave redundantly scans them in its first NW lanes (NW <= 16: one DPP row),
// so a block of up to 1024 threads needs a single barrier and no per-thread loop over the waves
// part 1 (before the barrier): per-wave inclusive scan, wave total to LDS; part 2 (after the barrier)
// finishes.  Split so that a caller can post other per-wave results under the same barrier.
__device__ __forceinline__ Tup4 wave_scan_tup4(Tup4 v, Tup4 *smem, Tup4 &exc) {
    const int lane = lane_id(), wv = wave_id();
    Tup4 inc = v;
    inc = tup4_combine(tup4_dpp<0x111, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x112, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x114, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x118, 0xf>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x142, 0xa>(inc), inc);
    inc = tup4_combine(tup4_dpp<0x143, 0xc>(inc), inc);
    if (lane == 63) smem[wv] = inc;
#pragma unroll
    for (int q = 0; q < 4; ++q) { exc.c[q] = lane_shr1(inc.c[q], 0); exc.t[q] = lane_shr1(inc.t[q], 0); }
    exc.all = lane_shr1(inc.all, 0);
    return inc;
}
template <int NW>
__device__ __forceinline__ Tup4 block_scan_finish_tup4(const Tup4 &exc, const Tup4 *smem, Tup4 &total) {
    const int lane = lane_id(), wv = wave_id();
    Tup4 wt = tup4_id();
    if (lane < NW) wt = smem[lane];
    wt = tup4_combine(tup4_dpp<0x111, 0xf>(wt), wt);
    wt = tup4_combine(tup4_dpp<0x112, 0xf>(wt), wt);
    if (NW > 4) { wt = tup4_combine(tup4_dpp<0x114, 0xf>(wt), wt); wt = tup4_combine(tup4_dpp<0x118, 0xf>(wt), wt); }
    Tup4 pre = tup4_id(), tot;
    const int src = (wv > 0) ? wv - 1 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pc = __builtin_amdgcn_readlane(wt.c[q], src), pt = __builtin_amdgcn_readlane(wt.t[q], src);
        if (wv > 0) { pre.c[q] = pc; pre.t[q] = pt; }
        tot.c[q] = __builtin_amdgcn_readlane(wt.c[q], NW - 1); tot.t[q] = __builtin_amdgcn_readlane(wt.t[q], NW - 1);
    }
    { const int pa = __builtin_amdgcn_readlane(wt.all, src); if (wv > 0) pre.all = pa; tot.all = __builtin_amdgcn_readlane(wt.all, NW - 1); }
    total = tot;
    return tup4_combine(pre, exc);
}

// combine one value per wave across the block (first NW lanes of every wave reduce the NW wave values)
template <int NW, bool IS_MAX>
__device__ __forceinline__ int waves_combine(const int *col /* stride 16 ints per wave */, int lane) {
    int v = (lane < NW) ? col[lane * 16] : 0;
    if (IS_MAX) {
        v = max(v, dpp_mov<0x111, 0xf>(0, v)); v = max(v, dpp_mov<0x112, 0xf>(0, v));
        if (NW > 4) { v = max(v, dpp_mov<0x114, 0xf>(0, v)); v = max(v, dpp_mov<0x118, 0xf>(0, v)); }
    } else {
        v += dpp_mov<0x111, 0xf>(0, v); v += dpp_mov<0x112, 0xf>(0, v);
        if (NW > 4) { v += dpp_mov<0x114, 0xf>(0, v); v += dpp_mov<0x118, 0xf>(0, v); }
    }
    return __builtin_amdgcn_readlane(v, NW - 1);
}

struct Step2Args {
    const int *a_in; const int *d_in;      // slot 2*jl   (state before site k = kbase + 2*jl)
    int *a_mid; int *d_mid;                // slot 2*jl+1 (before site k+1)
    int *a_out; int *d_out;                // slot 2*jl+2 (before site k+2)
    const Ctl *ctl;
    int4 *summ;                            // [3][wpad][3] int4: {c[4]}, {last[4]}, {maxd,0,0,0}
    long long *prof;
    int wpc, jl, M, W, wpad;
};

// NT threads per workgroup, E consecutive positions per thread: tile of T = NT*E positions
// (NT=256,E=1 for M <= 262144; NT=256,E=4 up to M = 1048576, all tiles resident at once).
template <bool WITH_D, bool FULL, int SPT, int NT, int E>
__device__ __forceinline__ void step2_body(const Step2Args &g, Tup4 *s_tup, int (*s_red)[16], int *s_acc) {
    constexpr int T = NT * E, NW = NT / 64;
    const int jl = g.jl;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T, i0 = S + t * E;
    PBWT_STAMP(0);
    const int4 *sm_in = g.summ + (size_t)(jl % 3) * g.wpad * 3;
    int4 *sm_out = g.summ + (size_t)((jl + 1) % 3) * g.wpad * 3;
    int4 *sm_zero = g.summ + (size_t)((jl + 2) % 3) * g.wpad * 3;
    if (t < 72) s_acc[t] = 0;

    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int a[E], d[E];
    if constexpr (E == 4) {
        const int4 va = *reinterpret_cast<const int4 *>(g.a_in + i0);
        a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
        if (WITH_D) { const int4 vd = *reinterpret_cast<const int4 *>(g.d_in + i0); d[0] = vd.x; d[1] = vd.y; d[2] = vd.z; d[3] = vd.w; }
        else { d[0] = d[1] = d[2] = d[3] = 0; }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { a[e] = g.a_in[i0 + e]; d[e] = WITH_D ? g.d_in[i0 + e] : 0; }
    }
    int4 sc[SPT], sl[SPT]; int smx[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * NT;
        sc[q] = make_int4(0, 0, 0, 0); sl[q] = make_int4(0, 0, 0, 0); smx[q] = 0;
        if (jn < W) { sc[q] = sm_in[(size_t)jn * 3]; if (WITH_D) { sl[q] = sm_in[(size_t)jn * 3 + 1]; smx[q] = sm_in[(size_t)jn * 3 + 2].x; } }
    }
    const int k = ctl.kbase + 2 * jl;
    // alleles of the owned haplotypes at sites k+2, k+3: the tags of slot 2*jl+2 = the next launch's keys
    const uint32_t *c2 = (k + 2 < ctl.n_total) ? ctl.cols + (size_t)(2 * jl + 2) * g.wpc : ctl.zerocol;
    const uint32_t *c3 = (k + 3 < ctl.n_total) ? ctl.cols + (size_t)(2 * jl + 3) * g.wpc : ctl.zerocol;
    int key[E], nkey[E];
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        valid[e] = FULL || (i0 + e < M);
        key[e] = (int)(((unsigned)a[e] >> 31) | (((unsigned)a[e] >> 29) & 2u));     // b0 | b1<<1
        a[e] &= AMASK;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        nkey[e] = 0;
        if (valid[e]) nkey[e] = (int)(((c2[(unsigned)a[e] >> 5] >> (a[e] & 31)) & 1u) | (((c3[(unsigned)a[e] >> 5] >> (a[e] & 31)) & 1u) << 1));
    }

    // ---- fold the tile summaries ----
    int bef[4] = {0, 0, 0, 0}, tot4[4] = {0, 0, 0, 0}, lst[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * NT;
        const int cc[4] = {sc[q].x, sc[q].y, sc[q].z, sc[q].w};
        const int ll[4] = {sl[q].x, sl[q].y, sl[q].z, sl[q].w};
#pragma unroll
        for (int x = 0; x < 4; ++x) { tot4[x] += cc[x]; if (jn < w) { bef[x] += cc[x]; lst[x] = max(lst[x], ll[x]); } }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) { bef[x] = wave_iscan_sum(bef[x]); tot4[x] = wave_iscan_sum(tot4[x]); if (WITH_D) lst[x] = wave_iscan_max(lst[x]); }
    if (lane == 63) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { s_red[wv][x] = bef[x]; s_red[wv][4 + x] = tot4[x]; s_red[wv][8 + x] = lst[x]; }
    }
    lds_barrier();
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        bef[x] = waves_combine<NW, false>(&s_red[0][x], lane);
        tot4[x] = waves_combine<NW, false>(&s_red[0][4 + x], lane);
        lst[x] = WITH_D ? waves_combine<NW, true>(&s_red[0][8 + x], lane) : 0;
    }
    PBWT_STAMP(1);
    // carries: max d_k over [last[x], S) = whole-tile maxima + one partial-tile read (<= T positions) per key
    int mx[4] = {0, 0, 0, 0};
    int pd[4][E];
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int tl = lst[x] ? (lst[x] - 1) / T : -1;
            const int hi = lst[x] ? min((tl + 1) * T, S) : 0;
#pragma unroll
            for (int e = 0; e < E; ++e) { const int p = lst[x] + t + e * NT; pd[x][e] = (p < hi) ? g.d_in[p] : 0; }
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int jn = t + q * NT; if (jn < w && jn > tl) mx[x] = max(mx[x], smx[q]); }
        }
    }

    // ---- the thread's own tuple (E positions in order), block scan ----
    Tup4 me = tup4_id();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (valid[e]) {
#pragma unroll
            for (int x = 0; x < 4; ++x) { if (x == key[e]) { ++me.c[x]; me.t[x] = 0; } else me.t[x] = max(me.t[x], d[e]); }
            me.all = max(me.all, d[e]);
        }
    }
    Tup4 tot, exc;
    wave_scan_tup4(me, s_tup, exc);
    PBWT_STAMP(2);
    // the carries' per-wave maxima ride on the scan's barrier (the partial-tile loads had the scan to land)
    int cr[4] = {0, 0, 0, 0};
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
#pragma unroll
            for (int e = 0; e < E; ++e) mx[x] = max(mx[x], pd[x][e]);
            mx[x] = wave_iscan_max(mx[x]);
        }
        if (lane == 63) {
#pragma unroll
            for (int x = 0; x < 4; ++x) s_red[wv][12 + x] = mx[x];
        }
    }
    lds_barrier();
    Tup4 run = block_scan_finish_tup4<NW>(exc, s_tup, tot);   // exclusive prefix of this thread's first position
    if (WITH_D) {
#pragma unroll
        for (int x = 0; x < 4; ++x) cr[x] = waves_combine<NW, true>(&s_red[0][12 + x], lane);
    }
    PBWT_STAMP(3);
    // ---- per position: divergences and destinations at both levels, scatter, next-pair summaries ----
    const int Zw1 = bef[0] + bef[2], C1 = tot4[0] + tot4[2];
    int G2[4]; G2[0] = 0; G2[1] = tot4[0]; G2[2] = tot4[0] + tot4[1]; G2[3] = tot4[0] + tot4[1] + tot4[2];
    const bool has_next = (k + 2 < ctl.n_total);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (valid[e]) {
            const int ky = key[e], de = d[e];
            int dd1 = 0, dd2 = 0;
            if (WITH_D) {
                int eff[4]; bool ex[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { ex[x] = run.c[x] || lst[x]; eff[x] = run.c[x] ? run.t[x] : max(cr[x], run.all); }
                int e2 = 0; bool x2 = false;
#pragma unroll
                for (int x = 0; x < 4; ++x) if (x == ky) { e2 = eff[x]; x2 = ex[x]; }
                if (x2) dd2 = max(e2, de);                 // same key: range max of d_k since that predecessor
                else {                                     // first of its key: differs from the last element of the nearest lower non-empty key
                    int lower = -1;
#pragma unroll
                    for (int x = 0; x < 4; ++x) if (x < ky && tot4[x] > 0) lower = x;
                    dd2 = (lower >= 0) ? k + 1 + (31 - __clz(ky ^ lower)) : 0;
                }
                // level 1: same allele at site k = the later of the two keys sharing b0 = the smaller maximum
                const int bb = ky & 1;
                const int ea = bb ? eff[1] : eff[0], eb = bb ? eff[3] : eff[2];
                const bool xa = bb ? ex[1] : ex[0], xb = bb ? ex[3] : ex[2];
                dd1 = (xa || xb) ? max(min(xa ? ea : 0x7fffffff, xb ? eb : 0x7fffffff), de) : k + 1;
            }
            const int b0 = ky & 1, b1 = ky >> 1;
            const int zr = run.c[0] + run.c[2], orr = run.c[1] + run.c[3];
            int prk = 0, base2 = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) if (x == ky) { base2 = G2[x] + bef[x]; prk = run.c[x]; }
            const int av1 = a[e] | (int)((unsigned)b1 << 31);
            const int av2 = a[e] | (int)(((unsigned)(nkey[e] & 1) << 31) | ((unsigned)(nkey[e] >> 1) << 30));
            {
                const int pos1 = b0 ? C1 + (S - Zw1) + orr : Zw1 + zr;
                const int pos2 = base2 + prk;
                g.a_mid[pos1] = av1;
                g.a_out[pos2] = av2;
                if (WITH_D) {
                    g.d_mid[pos1] = pos1 ? dd1 : k + 2;    // sentinels (pbwtCore.c:507)
                    if (pos2 == 0) dd2 = k + 3;
                    g.d_out[pos2] = dd2;
                }
                if (has_next) {                            // <= 2 destination tiles per key stream
                    const int slot = ky * 2 + (pos2 / T - base2 / T);
                    atomicAdd(&s_acc[slot * 9 + nkey[e]], 1);
                    if (WITH_D) { atomicMax(&s_acc[slot * 9 + 4 + nkey[e]], pos2 + 1); atomicMax(&s_acc[slot * 9 + 8], dd2); }
                }
            }
            if (E > 1) {                                   // advance the running prefix past this position
#pragma unroll
                for (int x = 0; x < 4; ++x) { if (x == ky) { ++run.c[x]; run.t[x] = 0; } else run.t[x] = max(run.t[x], de); }
                run.all = max(run.all, de);
            }
        }
    }
    if (WITH_D && w == W - 1 && t == 0) { g.d_mid[M] = k + 2; g.d_out[M] = k + 3; }
    PBWT_STAMP(4);
    if (has_next) {
        lds_barrier();
        if (t < 72) {
            const int slot = t / 9, f = t - slot * 9, kq = slot >> 1;
            const int fq = (G2[kq] + bef[kq]) / T;         // first destination tile of stream kq
            const int dt = fq + (slot & 1);
            const int v = s_acc[t];
            if (v && dt < W) {
                int *so = reinterpret_cast<int *>(sm_out + (size_t)dt * 3) + f;
                if (f < 4) atomicAdd(so, v); else atomicMax(so, v);
            }
        }
    }
    if (t < 3) sm_zero[(size_t)w * 3 + t] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(5);
    PBWT_STAMP(6);
}

template <bool WITH_D, int SPT, int NT, int E>
__global__ __launch_bounds__(NT) void step2_kernel(Step2Args g) {
    __shared__ Tup4 s_tup[NT / 64];
    __shared__ int s_red[NT / 64][16];
    __shared__ int s_acc[72];
    if ((int)(blockIdx.x + 1) * NT * E <= g.M) step2_body<WITH_D, true, SPT, NT, E>(g, s_tup, s_red, s_acc);
    else step2_body<WITH_D, false, SPT, NT, E>(g, s_tup, s_red, s_acc);
}

}  // namespace pbwtk
