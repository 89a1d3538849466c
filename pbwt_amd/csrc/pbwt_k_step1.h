// pbwt_k_step1.h — single-site chain (step1_kernel / step_kernel), prepare kernels, cursor init.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// first pair of a pass (or after an odd-length batch): both allele tags of slot 0 from columns k, k+1
// and the pair summaries from scratch; clears the accumulation buffer of the first launch
struct Prep2Args { int *a; const int *d; const uint32_t *col0; const uint32_t *col1; int4 *summ; int M, W, wpad, with_d, T; };
__global__ __launch_bounds__(BLOCK) void prepare2_kernel(Prep2Args g) {
    __shared__ int s_acc[9];
    const int t = threadIdx.x, w = blockIdx.x;
    if (t < 9) s_acc[t] = 0;
    __syncthreads();
    for (int i = w * g.T + t; i < min((w + 1) * g.T, g.M); i += BLOCK) {
        const int a = g.a[i] & AMASK;
        const unsigned b0 = (g.col0[(unsigned)a >> 5] >> (a & 31)) & 1u, b1 = (g.col1[(unsigned)a >> 5] >> (a & 31)) & 1u;
        g.a[i] = a | (int)((b0 << 31) | (b1 << 30));
        const int key = (int)(b0 | (b1 << 1));
        atomicAdd(&s_acc[key], 1);
        atomicMax(&s_acc[4 + key], i + 1);
        if (g.with_d) atomicMax(&s_acc[8], g.d[i]);
    }
    __syncthreads();
    if (t == 0) {
        g.summ[(size_t)w * 3] = make_int4(s_acc[0], s_acc[1], s_acc[2], s_acc[3]);
        g.summ[(size_t)w * 3 + 1] = make_int4(s_acc[4], s_acc[5], s_acc[6], s_acc[7]);
        g.summ[(size_t)w * 3 + 2] = make_int4(s_acc[8], 0, 0, 0);
        int4 *nxt = g.summ + (size_t)g.wpad * 3;
        nxt[(size_t)w * 3] = make_int4(0, 0, 0, 0); nxt[(size_t)w * 3 + 1] = make_int4(0, 0, 0, 0); nxt[(size_t)w * 3 + 2] = make_int4(0, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// step1_kernel: the E = 1 specialisation (T = 256, M <= 262144), written for the shortest
// instruction stream: SPT = ceil(W/256) summaries per thread, results scattered straight from
// registers (no LDS staging: the zeros of a wave go to one contiguous destination range, the ones to
// another), every wave posts its own next-site summaries with global atomics.  Three LDS barriers.
template <bool WITH_D, bool SORTED, bool FULL, int SPT>
__device__ __forceinline__ void step1_body(const StepArgs &g, int *s_a, int *s_d, Tup *s_tup, int (*s_red)[6], int (*s_acc)[4]) {
    constexpr int T = BLOCK;
    const int j = g.j;
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, W = g.W, M = g.M;
    const int S = w * T, i = S + t;
    PBWT_STAMP(0);
    const int4 *sm_in = g.summ + (size_t)(j % 3) * g.wpad;
    int4 *sm_out = g.summ + (size_t)((j + 1) % 3) * g.wpad;
    int4 *sm_zero = g.summ + (size_t)((j + 2) % 3) * g.wpad;

    if (t < 16) s_acc[t >> 2][t & 3] = 0;
    // ---- issue everything whose address is known now ----
    const Ctl ctl = *g.ctl;
    int a = g.a_in[i];                                     // padded to W*T
    int d = WITH_D ? g.d_in[i] : 0;
    int4 sv[SPT];                                          // the W tile summaries, SPT per thread
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        sv[q] = (jn < W) ? sm_in[jn] : make_int4(0, 0, 0, 0);
    }
    const int k = ctl.kbase + j;
    const bool has_next = (k + 1 < ctl.n_total);
    const uint32_t *col_next = ctl.cols + (size_t)(j + 1) * g.wpc;
    const bool valid = FULL || (i < M);
    const unsigned y = ((unsigned)a) >> 31;
    a &= AMASK;
    unsigned nbit = 0;
    if (!SORTED && has_next && valid) nbit = (col_next[(unsigned)a >> 5] >> (a & 31)) & 1u;

    // ---- fold the summaries: zeros before the tile, zeros in the column, last 0 / 1 before the tile ----
    int sumBefore = 0, total = 0, l0 = 0, l1 = 0;
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int jn = t + q * BLOCK;
        total += sv[q].x;
        if (jn < w) { sumBefore += sv[q].x; if (WITH_D) { l0 = max(l0, sv[q].y); l1 = max(l1, sv[q].z); } }
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
    int m0 = 0, m1 = 0, pd0 = 0, pd1 = 0;
    if (WITH_D) {
        // carry_b = max d over [l_b, S): whole-tile maxima + one partial-tile read (<= 256 positions)
        const int tl0 = l0 ? (l0 - 1) / T : -1, tl1 = l1 ? (l1 - 1) / T : -1;
        const int hi0 = l0 ? min((tl0 + 1) * T, S) : 0, hi1 = l1 ? min((tl1 + 1) * T, S) : 0;
        if (l0 + t < hi0) pd0 = g.d_in[l0 + t];            // the one dependent load; consumed after the scan
        if (l1 + t < hi1) pd1 = g.d_in[l1 + t];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int jn = t + q * BLOCK;
            if (jn < w) { if (jn > tl0) m0 = max(m0, sv[q].w); if (jn > tl1) m1 = max(m1, sv[q].w); }
        }
    }
    PBWT_STAMP(1);

    // ---- the position's own tuple, block scan ----
    Tup me = Tup{0, 0, 0, 0, 0};
    if (valid) { if (y) { me.c1 = 1; me.t0 = d; } else { me.c0 = 1; me.t1 = d; } me.all = d; }
    Tup tot;
    const Tup pre = block_scan_tup<WITH_D>(me, s_tup, tot);
    PBWT_STAMP(2);
    int dn = 0;
    if (WITH_D) {
        m0 = wave_max(max(m0, pd0)); m1 = wave_max(max(m1, pd1));
        if (lane == 0) { s_red[wv][4] = m0; s_red[wv][5] = m1; }
        lds_barrier();
        m0 = 0; m1 = 0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { m0 = max(m0, s_red[q][4]); m1 = max(m1, s_red[q][5]); }
        const int carry0 = l0 ? m0 : k + 1;                // nothing before: p starts at k+1 (pbwtCore.c:489)
        const int carry1 = l1 ? m1 : k + 1;
        const int pin = y ? (pre.c1 ? pre.t1 : max(carry1, pre.all)) : (pre.c0 ? pre.t0 : max(carry0, pre.all));
        dn = max(pin, d);
    }
    PBWT_STAMP(3);
    // ---- stage in LDS in destination order (coalesced stores drain faster at kernel end), then
    //      write out + summaries of site k+1 ----
    const int cw = tot.c0, nvalid = tot.c0 + tot.c1;
    if (valid) {
        const int ldst = y ? cw + pre.c1 : pre.c0;
        s_a[ldst] = a | (int)(nbit << 31);
        if (WITH_D) s_d[ldst] = dn;
    }
    lds_barrier();
    const int oneBase = C + (S - Zw);                      // every earlier tile is full
    const int tz = Zw / T, to = oneBase / T;
    const bool ovalid = FULL || (t < nvalid);
    const bool one = t >= cw;
    const int P = one ? oneBase + (t - cw) : Zw + t;
    const int slot = ovalid ? (one ? 2 + (P / T - to) : (P / T - tz)) : -1;
    unsigned tag = 0;
    if (ovalid) {
        int ao = s_a[t];
        if (SORTED) { if (has_next) tag = (col_next[(unsigned)P >> 5] >> (P & 31)) & 1u; ao |= (int)(tag << 31); }
        else tag = (unsigned)ao >> 31;
        g.a_out[P] = ao;
        if (WITH_D) { dn = s_d[t]; if (P == 0) dn = k + 2; g.d_out[P] = dn; }      // sentinel (pbwtCore.c:507)
    }
    PBWT_STAMP(4);
    if (has_next) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned long long mk = __ballot(slot == s);
            if (mk) {                                      // wave-uniform
                const unsigned long long ones = __ballot(slot == s && tag);
                const unsigned long long zeros = mk & ~ones;
                // within a stream P grows with the lane: the highest lane of a set holds its last position
                const int pz = zeros ? __builtin_amdgcn_readlane(P, 63 - __clzll(zeros)) + 1 : 0;
                const int po = ones ? __builtin_amdgcn_readlane(P, 63 - __clzll(ones)) + 1 : 0;
                int md = 0;
                if (WITH_D) md = wave_max((slot == s) ? dn : 0);
                if (lane == 0) {                           // aggregate in LDS: 16 global atomics per tile, not per wave
                    if (zeros) atomicAdd(&s_acc[s][0], __popcll(zeros));
                    if (WITH_D) {
                        if (pz) atomicMax(&s_acc[s][1], pz);
                        if (po) atomicMax(&s_acc[s][2], po);
                        if (md) atomicMax(&s_acc[s][3], md);
                    }
                }
            }
        }
    }
    if (WITH_D && w == W - 1 && t == 0) g.d_out[M] = k + 2;
    if (t == 0) sm_zero[w] = make_int4(0, 0, 0, 0);
    PBWT_STAMP(5);
    if (has_next) {
        lds_barrier();
        if (t < 16) {
            const int s = t >> 2, f = t & 3;
            const int dt = (s < 2 ? tz : to) + (s & 1);
            const int v = s_acc[s][f];
            if (v && dt < W) {
                int *so = reinterpret_cast<int *>(sm_out + dt) + f;
                if (f == 0) atomicAdd(so, v); else atomicMax(so, v);
            }
        }
    }
    PBWT_STAMP(6);
}

template <bool WITH_D, bool SORTED, int SPT>
__global__ __launch_bounds__(BLOCK) void step1_kernel(StepArgs g) {
    __shared__ Tup s_tup[WAVES];
    __shared__ int s_red[WAVES][6];
    __shared__ int s_acc[4][4];
    __shared__ int s_a[BLOCK];
    __shared__ int s_d[WITH_D ? BLOCK : 1];
    if ((int)(blockIdx.x + 1) * BLOCK <= g.M) step1_body<WITH_D, SORTED, true, SPT>(g, s_a, s_d, s_tup, s_red, s_acc);
    else step1_body<WITH_D, SORTED, false, SPT>(g, s_a, s_d, s_tup, s_red, s_acc);
}

// One site of pbwtCursorForwardsA / ForwardsAD (pbwtCore.c:458-470 / 485-508) for one tile of
// T = 256*E consecutive positions.  grid = W tiles.
template <int E, bool WITH_D, bool SORTED>
__global__ __launch_bounds__(BLOCK) void step_kernel(StepArgs g) {
    constexpr int T = BLOCK * E;
    __shared__ int s_a[T];
    __shared__ int s_d[WITH_D ? T : 1];
    __shared__ Tup s_tup[WAVES];
    __shared__ int s_red[WAVES][6];
    __shared__ int s_acc[4][4];
    if ((int)(blockIdx.x + 1) * T <= g.M) step_body<E, WITH_D, SORTED, true>(g, s_a, s_d, s_tup, s_red, s_acc);
    else step_body<E, WITH_D, SORTED, false>(g, s_a, s_d, s_tup, s_red, s_acc);
}

// ---------------------------------------------------------------------------------------------
// prepare: first site of a pass.  Tags a[i] with y_k[i] from column k and builds that site's tile
// summaries from scratch (plain stores), zeroing the accumulation target of the first step.
struct PrepArgs {
    int *a; const int *d; const uint32_t *col; int4 *summ;
    int k, M, W, wpad, T, sorted, with_d, has_col;
};

__global__ __launch_bounds__(BLOCK) void prepare_kernel(PrepArgs g) {
    __shared__ int s_red[WAVES][4];
    const int t = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int w = blockIdx.x, S = w * g.T;
    int c0 = 0, l0 = 0, l1 = 0, md = 0;
    for (int l = t; l < g.T; l += BLOCK) {
        const int i = S + l;
        if (i < g.M) {
            const int a = g.a[i] & AMASK;
            unsigned y = 0;
            if (g.has_col) {
                const unsigned idx = g.sorted ? (unsigned)i : (unsigned)a;
                y = (g.col[idx >> 5] >> (idx & 31)) & 1u;
            }
            g.a[i] = a | (int)(y << 31);
            if (y == 0) { ++c0; l0 = max(l0, i + 1); } else l1 = max(l1, i + 1);
            if (g.with_d) md = max(md, g.d[i]);
        }
    }
    c0 = wave_sum(c0); l0 = wave_max(l0); l1 = wave_max(l1); md = wave_max(md);
    if (lane == 0) { s_red[wv][0] = c0; s_red[wv][1] = l0; s_red[wv][2] = l1; s_red[wv][3] = md; }
    __syncthreads();
    if (t == 0) {
        c0 = 0; l0 = 0; l1 = 0; md = 0;
        for (int q = 0; q < WAVES; ++q) { c0 += s_red[q][0]; l0 = max(l0, s_red[q][1]); l1 = max(l1, s_red[q][2]); md = max(md, s_red[q][3]); }
        g.summ[w] = make_int4(c0, l0, l1, md);               // batch-relative: step 0 reads buffer 0
        g.summ[(size_t)g.wpad + w] = make_int4(0, 0, 0, 0);
    }
}

// cursor init (pbwtNakedCursorCreate, pbwtCore.c:402-418): a = identity unless given; d = 0 with
// sentinels d[0] = d[M] = k0+1
__global__ void init_state_kernel(int *a, int *d, int M, int Mpad, int k0, int identity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Mpad) {
        if (identity) a[i] = (i < M) ? i : 0;
        else if (i >= M) a[i] = 0;
    }
    if (i <= Mpad) { if (d) d[i] = (i == 0 || i == M) ? k0 + 1 : 0; }
}

}  // namespace pbwtk
