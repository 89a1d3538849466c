// pbwt_k_common.h — wave64 cross-lane primitives on DPP, the carry tuple of the divergence recurrence and its block scan.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int WAVES = BLOCK / 64;
constexpr unsigned TAG = 0x80000000u;
constexpr int AMASK = 0x3fffffff;    // bits 31/30 of a ring entry carry the alleles at the slot's site / the next site

// workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory
// (s_waitcnt vmcnt(0)), which would serialise every barrier behind the outstanding global loads
// and stores this latency-bound kernel deliberately keeps in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// the same as an SGPR (wave-uniform by construction — every kernel here runs 1-D blocks of whole waves — but the compiler cannot know): every tile /
// slot index and pointer derived from it becomes scalar arithmetic.  Pays where a wave owns its work item (skel_fillseq_kernel: 89 -> 80 VGPRs);
// measured worse on the chain's hist / rank kernels (scalar branches on the wave index duplicate code: 28 -> 44 VGPRs), which keep wave_id().
__device__ __forceinline__ int wave_id_s() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- wave64 cross-lane primitives on DPP (row_shr + row_bcast15/31): a 6-op dependent chain of
// VALU instructions instead of 6 ds_bpermute round trips through the LDS crossbar.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xf, false);
}
#define PBWT_DPP_SCAN(v, OP, ID)                                      \
    v = OP(v, dpp_mov<0x111, 0xf>(ID, v)); /* row_shr:1 */            \
    v = OP(v, dpp_mov<0x112, 0xf>(ID, v)); /* row_shr:2 */            \
    v = OP(v, dpp_mov<0x114, 0xf>(ID, v)); /* row_shr:4 */            \
    v = OP(v, dpp_mov<0x118, 0xf>(ID, v)); /* row_shr:8 */            \
    v = OP(v, dpp_mov<0x142, 0xa>(ID, v)); /* row_bcast:15 */         \
    v = OP(v, dpp_mov<0x143, 0xc>(ID, v)); /* row_bcast:31 */
__device__ __forceinline__ int op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int op_max(int a, int b) { return max(a, b); }
__device__ __forceinline__ int wave_iscan_sum(int v) { PBWT_DPP_SCAN(v, op_add, 0) return v; }
__device__ __forceinline__ int wave_iscan_max(int v) { PBWT_DPP_SCAN(v, op_max, 0) return v; }   // values >= 0
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_iscan_sum(v), 63); }
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_iscan_max(v), 63); }
// value of the previous lane (lane 0 gets `id`)
__device__ __forceinline__ int lane_shr1(int v, int id) {
    return __builtin_amdgcn_update_dpp(id, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// ---- the 16-bit hand-off between the fill and the -stats sweep (round 4; skel_fillseq_kernel<.., 3> -> sweep_hist_kernel<true, false, true>).
// Slot s of a batch, position i: L | y << 15 with L = (site of the slot) + 1 - d[i] (the match length with the neighbour above, plus one; 0 for the
// sentinels) or P16_ESC when L >= clip (clip = P16_ESC; smaller only in tests): then, and only then, d[i] itself stands in the 32-bit ring slot
// and the reader fetches it from there.
constexpr int P16_ESC = 0x7fff;
__device__ __forceinline__ unsigned p16_encode(int site, int d, unsigned y, int clip, bool &esc) {
    const int L = site + 1 - d;
    esc = L >= clip;
    return (unsigned)(esc ? P16_ESC : L) | (y << 15);
}
// a word of the 16-bit ring as the d | y << 31 word the sweeps work on (kp1 = the slot's site + 1; x = the position, for the rare escape)
__device__ __forceinline__ int p16_word(unsigned h, int kp1, const int *d, int x) {
    const int L = (int)(h & 0x7fffu);
    const int dv = (L == P16_ESC) ? (__builtin_nontemporal_load(d + x) & 0x7fffffff) : kp1 - L;
    return dv | (int)((h >> 15) << 31);
}

__device__ __forceinline__ uint64_t sm64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------
// carry tuple of the divergence recurrence (pbwtCore.c:492-503).  For a segment of positions:
//   c0,c1 = number of 0 / 1 alleles; all = max d over the segment;
//   t_b   = max d over the elements after the last allele-b element (all if there is none).
// combine(L,R) is associative; (0,0,0,0,0) is the identity (d >= 0 everywhere).
struct Tup { int c0, c1, t0, t1, all; };

__device__ __forceinline__ Tup tup_combine(const Tup &L, const Tup &R) {
    Tup o;
    o.c0 = L.c0 + R.c0;
    o.c1 = L.c1 + R.c1;
    o.all = max(L.all, R.all);
    o.t0 = R.c0 ? R.t0 : max(L.t0, R.all);
    o.t1 = R.c1 ? R.t1 : max(L.t1, R.all);
    return o;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ Tup tup_dpp(const Tup &v) {      // unwritten lanes get the identity
    Tup r;
    r.c0 = dpp_mov<CTRL, ROWMASK>(0, v.c0); r.c1 = dpp_mov<CTRL, ROWMASK>(0, v.c1);
    r.t0 = dpp_mov<CTRL, ROWMASK>(0, v.t0); r.t1 = dpp_mov<CTRL, ROWMASK>(0, v.t1);
    r.all = dpp_mov<CTRL, ROWMASK>(0, v.all);
    return r;
}
template <bool WITH_D>
__device__ __forceinline__ Tup tup_op(const Tup &L, const Tup &R) {
    if (WITH_D) return tup_combine(L, R);
    return Tup{L.c0 + R.c0, L.c1 + R.c1, 0, 0, 0};
}

// block-wide exclusive scan of Tup over 256 threads (lane order = position order); also returns
// the block total.  smem: WAVES Tups.  One __syncthreads.
template <bool WITH_D>
__device__ __forceinline__ Tup block_scan_tup(Tup v, Tup *smem, Tup &total) {
    const int lane = lane_id(), wv = wave_id();
    Tup inc = v;
    inc = tup_op<WITH_D>(tup_dpp<0x111, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x112, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x114, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x118, 0xf>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x142, 0xa>(inc), inc);
    inc = tup_op<WITH_D>(tup_dpp<0x143, 0xc>(inc), inc);
    if (lane == 63) smem[wv] = inc;
    Tup exc;
    exc.c0 = lane_shr1(inc.c0, 0); exc.c1 = lane_shr1(inc.c1, 0);
    exc.t0 = lane_shr1(inc.t0, 0); exc.t1 = lane_shr1(inc.t1, 0); exc.all = lane_shr1(inc.all, 0);
    lds_barrier();
    Tup pre = Tup{0, 0, 0, 0, 0};
    Tup tot = Tup{0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const Tup sw = smem[w];
        if (w < wv) pre = tup_op<WITH_D>(pre, sw);
        tot = tup_op<WITH_D>(tot, sw);
    }
    total = tot;
    return tup_op<WITH_D>(pre, exc);
}

}  // namespace pbwtk
