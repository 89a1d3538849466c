// pbwt_k_codec.h — pack3 codec (pbwtCore.c:240-305): 64-bit scans, region-parallel encoder, decoder with validation.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// single-block exclusive scan of n 64-bit values (in place), total to *total
__global__ __launch_bounds__(1024) void scan_u64_kernel(unsigned long long *v, size_t n, unsigned long long *total,
                                                       unsigned long long base_in) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = base_in;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t b = 0; b < n; b += 1024) {
        const size_t i = b + threadIdx.x;
        const unsigned long long x = (i < n) ? v[i] : 0ULL;
        unsigned long long inc = x;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) pre += s_w[q]; tot += s_w[q]; }
        if (i < n) v[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = s_carry;
}

// large arrays: per-block sums (SCAN_CHUNK values per workgroup) -> scan_u64_kernel over the block sums -> local
// exclusive scan + block offset.  (the single-block kernel streams at one workgroup's bandwidth: 5 ms for 10 M values)
constexpr int SCAN_CHUNK = 4096;
__global__ __launch_bounds__(BLOCK) void scan_u64_blocksum_kernel(const unsigned long long *v, size_t n, unsigned long long *bsum) {
    __shared__ unsigned long long s_w[WAVES];
    const size_t b0 = (size_t)blockIdx.x * SCAN_CHUNK;
    unsigned long long acc = 0;
    for (int x = threadIdx.x; x < SCAN_CHUNK; x += BLOCK) { const size_t i = b0 + x; if (i < n) acc += v[i]; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane_id() == 0) s_w[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int q = 0; q < WAVES; ++q) t += s_w[q]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(BLOCK) void scan_u64_apply_kernel(unsigned long long *v, size_t n, const unsigned long long *boff) {
    __shared__ unsigned long long s_w[WAVES];
    __shared__ unsigned long long s_carry;
    const size_t b0 = (size_t)blockIdx.x * SCAN_CHUNK;
    const int lane = lane_id(), wv = wave_id();
    if (threadIdx.x == 0) s_carry = boff[blockIdx.x];
    __syncthreads();
    for (int x0 = 0; x0 < SCAN_CHUNK; x0 += BLOCK) {
        const size_t i = b0 + x0 + threadIdx.x;
        const unsigned long long x = (i < n) ? v[i] : 0ULL;
        unsigned long long inc = x;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wv) pre += s_w[q]; tot += s_w[q]; }
        if (i < n) v[i] = pre + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// sorted bit columns out of the ring tags: ycol[site][word] (one wave per 64 positions)
__global__ __launch_bounds__(BLOCK) void tags_to_bits_kernel(const int *A, size_t strideA, int M, unsigned long long *ycols,
                                                            int wpc64) {
    const int site = blockIdx.y;
    const int *a = A + (size_t)site * strideA;
    const int nw = (M + 63) / 64;
    for (int wd = blockIdx.x * WAVES + wave_id(); wd < wpc64; wd += gridDim.x * WAVES) {
        const int i = wd * 64 + lane_id();
        const bool one = (wd < nw) && (i < M) && (a[i] < 0);
        const unsigned long long mk = __ballot(one);
        if (lane_id() == 0) ycols[(size_t)site * wpc64 + wd] = mk;
    }
}

// ---------------------------------------------------------------------------------------------
// pack3 encode (pbwtCore.c:240-267) of sorted bit columns.  One block per column; each thread
// owns 64-position words; a run is emitted by the word in which it ENDS.
// bytes for a run of length n (pack3Add, pbwtCore.c:240-252)
__device__ __forceinline__ int p3_nbytes(int n) {
    int c = 0;
    if (n >= 63488) { c = n / 63488; n -= c * 63488; }       // rare: keep the division off the common path
    if (n >= 2048) { ++c; n &= 0x7ff; }
    if (n >= 64) { ++c; n &= 0x3f; }
    if (n) ++c;
    return c;
}
__device__ __forceinline__ uint8_t *p3_emit(uint8_t *o, unsigned v, int n) {
    const uint8_t top = (uint8_t)(v << 7);
    while (n >= 63488) { *o++ = top | 0x7f; n -= 63488; }
    if (n >= 2048) { *o++ = top | 0x60 | (uint8_t)(n >> 11); n &= 0x7ff; }
    if (n >= 64) { *o++ = top | 0x40 | (uint8_t)(n >> 6); n &= 0x3f; }
    if (n) *o++ = top | (uint8_t)n;
    return o;
}

#ifdef PBWTAMD_MEASURE   // the two earlier encoders: measurement builds only (A/B runs against the region-parallel form below)
// MODE 0: colBytes[col] = encoded size; MODE 1: write bytes at colOffset[col]
// NT threads per column: the loop over chunks of NT words is a chain of barriers and dependent loads (latency of ONE
// workgroup, whatever the batch), so wide columns take 1024 threads.
template <int MODE, int NT = BLOCK>
__global__ __launch_bounds__(NT) void pack3_kernel(const unsigned long long *ycols, int wpc64, int M,
                                                     unsigned long long *colBytes, uint8_t *out) {
    constexpr int NWV = NT / 64;
    __shared__ int s_wi[NWV];
    __shared__ int s_carry_start;       // start position of the run open at the chunk boundary
    __shared__ int s_carry_bytes;       // bytes emitted so far in this column
    const int col = blockIdx.x;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64;
    const int lane = lane_id(), wv = wave_id();
    if (threadIdx.x == 0) { s_carry_start = 0; s_carry_bytes = 0; }
    __syncthreads();
    uint8_t *obase = (MODE == 1) ? out + colBytes[col] : nullptr;
    for (int b = 0; b < nw; b += NT) {
        const int wd = b + threadIdx.x;
        unsigned long long cur = 0, trans = 0;
        int nbits = 0;
        if (wd < nw) {
            cur = y[wd];
            nbits = min(64, M - wd * 64);
            const unsigned long long prevbit = (wd > 0) ? (y[wd - 1] >> 63) : 0ULL;
            trans = cur ^ ((cur << 1) | prevbit);          // bit p set: position starts a new run
            if (wd == 0) trans &= ~1ULL;                   // position 0 opens the first run, closes nothing
            if (nbits < 64) trans &= (1ULL << nbits) - 1ULL;
        }
        // last run start at or before the beginning of this word: max-scan of last transition pos
        int lastT = trans ? (wd * 64 + 63 - __clzll(trans)) : -1;
        int incl = lastT;
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl = max(incl, v); }
        if (lane == 63) s_wi[wv] = incl;
        int exclT = __shfl_up(incl, 1); if (lane == 0) exclT = -1;
        __syncthreads();
        int preT = -1;
        for (int q = 0; q < NWV; ++q) if (q < wv) preT = max(preT, s_wi[q]);
        int chunkLast = -1;
        for (int q = 0; q < NWV; ++q) chunkLast = max(chunkLast, s_wi[q]);
        int open = max(max(exclT, preT), -1);
        if (open < 0) open = s_carry_start;                // run opened in an earlier chunk (or at 0)
        // runs closed by this word: one per transition, plus the final run if this word holds M-1
        const bool lastWord = (wd == nw - 1);
        int myBytes = 0;
        {
            unsigned long long tr = trans; int st = open;
            while (tr) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; tr &= tr - 1; myBytes += p3_nbytes(pz - st); st = pz; }
            if (lastWord) myBytes += p3_nbytes(M - st);
        }
        // exclusive scan of myBytes within the chunk
        int inc = myBytes;
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        __syncthreads();
        if (lane == 63) s_wi[wv] = inc;
        __syncthreads();
        int preB = s_carry_bytes, totB = 0;
        for (int q = 0; q < NWV; ++q) { if (q < wv) preB += s_wi[q]; totB += s_wi[q]; }
        if (MODE == 1 && myBytes) {
            uint8_t *o = obase + preB + inc - myBytes;
            unsigned long long tr = trans; int st = open;
            while (tr) {
                const int pz = wd * 64 + __ffsll((long long)tr) - 1; tr &= tr - 1;
                // value of the run [st,pz) = bit at st
                const unsigned v = (unsigned)((y[st >> 6] >> (st & 63)) & 1ULL);
                o = p3_emit(o, v, pz - st); st = pz;
            }
            if (lastWord) { const unsigned v = (unsigned)((y[st >> 6] >> (st & 63)) & 1ULL); o = p3_emit(o, v, M - st); }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry_bytes += totB; if (chunkLast >= 0) s_carry_start = chunkLast; }
        __syncthreads();
    }
    if (MODE == 0 && threadIdx.x == 0) colBytes[col] = (unsigned long long)s_carry_bytes;
}

// pack3 encode, wave-regional: a wave owns a contiguous region of 64*IT words of the column and walks it 64 words
// (= one coalesced 512-byte load) at a time; all IT loads are issued up front.  Inside the wave the start of the run open
// at a word is an exclusive max-scan over the lanes (DPP) carried across the iterations — no barrier; across the waves of
// the column ONE LDS exchange of (first / last transition, bytes) fixes the run open at each region's start and the byte
// bases.  A run is emitted by the word in which it ends; its value is the last bit of the previous word and alternates from
// there.  (pack3_kernel above does the same with a barrier chain per 1024-word chunk: 16 chunks x 5 barriers at M = 1 M.)
template <int MODE, int NT, int IT>
__global__ __launch_bounds__(NT) void pack3v2_kernel(const unsigned long long *ycols, int wpc64, int M,
                                                       unsigned long long *colBytes, uint8_t *out) {
    constexpr int NWV = NT / 64;
    __shared__ int s_last[NWV], s_first[NWV], s_inner[NWV];
    const int col = blockIdx.x, lane = lane_id(), wv = wave_id();
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = wv * 64 * IT;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    const int hi0 = (base > 0 && base <= nw) ? (int)(y[base - 1] >> 32) : 0;      // the word before the region (its last bit matters)
    // transitions of word i of this lane: bit p set = position 64 wd + p starts a new run.  prevHi carries the previous
    // iteration's last word across the loop.
    auto transitions = [&](int i, int &prevHi) -> unsigned long long {
        const int wd = base + i * 64 + lane;
        const int hi = (int)(cur[i] >> 32);
        const int ph = lane_shr1(hi, prevHi);
        prevHi = __builtin_amdgcn_readlane(hi, 63);
        unsigned long long tr = cur[i] ^ ((cur[i] << 1) | (unsigned long long)((unsigned)ph >> 31));
        if (wd == 0) tr &= ~1ULL;                            // position 0 opens the first run, closes nothing
        const int nbits = M - wd * 64;
        if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
        return tr;
    };
    // ---- pass A: first / last transition of the region, bytes of the runs that start at a transition of the region and end in it
    int carryT = -1, firstT = -1, inner = 0, prevHi = hi0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        unsigned long long tr = transitions(i, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1, tf = tr ? wd * 64 + __ffsll((long long)tr) - 1 : -1;
        const int inc = wave_iscan_max(tl + 1);             // 1 + last transition up to and including this lane (0 = none)
        int st = max(lane_shr1(inc, 0) - 1, carryT);        // start of the run open at this word; -1 = it began before the region
        int bytes = 0;
        for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; if (st >= 0) bytes += p3_nbytes(pz - st); st = pz; }
        inner += wave_sum(bytes);
        const unsigned long long has = __ballot(tf >= 0);
        if (has) {
            if (firstT < 0) firstT = __builtin_amdgcn_readlane(tf, __ffsll((long long)has) - 1);
            carryT = __builtin_amdgcn_readlane(inc, 63) - 1;
        }
    }
    if (lane == 0) { s_last[wv] = carryT; s_first[wv] = firstT; s_inner[wv] = inner; }
    __syncthreads();
    // ---- the waves before this one: run open at the region's start, byte base
    int openW = 0, baseB = 0, total = 0;
    {
        int open = 0;                                        // start of the run open at wave q's region (position 0 opens the first run)
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            const int lq = s_last[q], fq = s_first[q];
            const bool ownsLast = (q * 64 * IT < nw) && ((q + 1) * 64 * IT >= nw);
            const int wb = (fq >= 0 ? p3_nbytes(fq - open) : 0) + s_inner[q] + (ownsLast ? p3_nbytes(M - (lq >= 0 ? lq : open)) : 0);
            if (q == wv) { openW = open; baseB = total; }
            total += wb;
            if (lq >= 0) open = lq;
        }
    }
    if (MODE == 0) { if (threadIdx.x == 0) colBytes[col] = (unsigned long long)total; return; }
    // ---- pass B: emit.  The run open at the region's start now has a known start (openW).
    // Scattered single-byte stores to HBM are slow (partial-sector writes): a column of up to P3_STAGE bytes — all but iid-like
    // columns of wide panels — is assembled in LDS and copied out with consecutive lanes writing consecutive bytes.
    constexpr int P3_STAGE = 32768;
    __shared__ uint8_t s_stage[MODE == 1 ? P3_STAGE : 1];
    const bool staged = total <= P3_STAGE;
    uint8_t *obase = (staged ? s_stage : out + colBytes[col]) + baseB;
    // (a rolled loop over freshly reloaded, L2-hot words: keeping all IT words live through the emission code spills registers)
    carryT = openW; prevHi = hi0;
    int done = 0;                                            // bytes emitted so far by this wave
    unsigned long long nxt = (base + lane < nw) ? y[base + lane] : 0ULL;
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        if (base + i * 64 >= nw) break;
        const unsigned long long cw = nxt;
        nxt = (wd + 64 < nw && i + 1 < IT) ? y[wd + 64] : 0ULL;
        const int hiPrevIter = prevHi;                       // the value of the run open at this word = the last bit before it
        unsigned long long tr;
        {
            const int hi = (int)(cw >> 32);
            const int ph0 = lane_shr1(hi, prevHi);
            prevHi = __builtin_amdgcn_readlane(hi, 63);
            tr = cw ^ ((cw << 1) | (unsigned long long)((unsigned)ph0 >> 31));
            if (wd == 0) tr &= ~1ULL;
            const int nbits = M - wd * 64;
            if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
        }
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1;
        const int inc = wave_iscan_max(tl + 1);
        int st = max(lane_shr1(inc, 0) - 1, carryT);
        const bool lastWord = (wd == nw - 1);
        int bytes = 0;
        { int s2 = st; for (unsigned long long t2 = tr; t2; t2 &= t2 - 1) { const int pz = wd * 64 + __ffsll((long long)t2) - 1; bytes += p3_nbytes(pz - s2); s2 = pz; } if (lastWord) bytes += p3_nbytes(M - s2); }
        const int incB = wave_iscan_sum(bytes);
        const int ph = lane_shr1((int)(cw >> 32), hiPrevIter);   // cross-lane: outside the divergent branch below
        if (bytes) {
            uint8_t *o = obase + done + incB - bytes;
            unsigned v = (wd == 0) ? (unsigned)(cw & 1ULL) : ((unsigned)ph >> 31);
            for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; o = p3_emit(o, v, pz - st); st = pz; v ^= 1u; }
            if (lastWord) p3_emit(o, v, M - st);
        }
        done += __builtin_amdgcn_readlane(incB, 63);
        const int wl = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (wl >= 0) carryT = wl;
    }
    if (staged) {
        __syncthreads();
        uint8_t *dst = out + colBytes[col];
        for (int x = threadIdx.x; x < total; x += NT) dst[x] = s_stage[x];
    }
}
#endif  // PBWTAMD_MEASURE

// pack3 encode, region-parallel (three launches per batch of columns): a column is cut into REGIONS of 64*IT words, one wave
// each, and no wave waits for another.
//   p3r_scan_kernel    per region: first / last transition, bytes of the runs that start and end inside it
//   p3r_combine_kernel per column (one wave): the run open at each region's start (max-scan over the regions' last transitions),
//                      the regions' byte bases (sum-scan), the column's size
//   p3r_emit_kernel    per region: emission at the column's offset + the region's base
// pack3v2_kernel does the same inside one workgroup per column; its 16 waves x 16 serial iterations at M = 1 M are a latency
// chain (0.27 ms per 512 columns) where this form runs 245 single-iteration waves per column.
struct P3Region { int firstT, lastT, inner, pad; };           // after combine: {openW, baseB, -, -}

template <int IT>
__device__ __forceinline__ unsigned long long p3r_transitions(const unsigned long long (&cur)[IT], int i, int base, int lane, int M, int &prevHi) {
    const int wd = base + i * 64 + lane;
    const int hi = (int)(cur[i] >> 32);
    const int ph = lane_shr1(hi, prevHi);                    // previous word's high half (lane 0: the last word before this iteration)
    prevHi = __builtin_amdgcn_readlane(hi, 63);
    unsigned long long tr = cur[i] ^ ((cur[i] << 1) | (unsigned long long)((unsigned)ph >> 31));
    if (wd == 0) tr &= ~1ULL;                                // position 0 opens the first run, closes nothing
    const int nbits = M - wd * 64;
    if (nbits <= 0) tr = 0; else if (nbits < 64) tr &= (1ULL << nbits) - 1ULL;
    return tr;
}

template <int IT>
__global__ __launch_bounds__(BLOCK) void p3r_scan_kernel(const unsigned long long *ycols, int wpc64, int M, int R, P3Region *regs) {
    const int col = blockIdx.y, reg = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (reg >= R) return;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = reg * 64 * IT;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    int prevHi = (base > 0 && base <= nw) ? (int)(y[base - 1] >> 32) : 0;
    int carryT = -1, firstT = -1, inner = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        unsigned long long tr = p3r_transitions<IT>(cur, i, base, lane, M, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1, tf = tr ? wd * 64 + __ffsll((long long)tr) - 1 : -1;
        const int inc = wave_iscan_max(tl + 1);             // 1 + last transition up to and including this lane (0 = none)
        int st = max(lane_shr1(inc, 0) - 1, carryT);        // start of the run open at this word; -1 = it began before the region
        int bytes = 0;
        for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; if (st >= 0) bytes += p3_nbytes(pz - st); st = pz; }
        inner += wave_sum(bytes);
        const unsigned long long has = __ballot(tf >= 0);
        if (has) {
            if (firstT < 0) firstT = __builtin_amdgcn_readlane(tf, __ffsll((long long)has) - 1);
            carryT = __builtin_amdgcn_readlane(inc, 63) - 1;
        }
    }
    if (lane == 0) regs[(size_t)col * R + reg] = P3Region{firstT, carryT, inner, 0};
}

// one wave per column: lanes = regions, 64 at a time with carries
__global__ __launch_bounds__(BLOCK) void p3r_combine_kernel(int M, int R, int words_per_region, int ncols, P3Region *regs, unsigned long long *colBytes) {
    const int col = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (col >= ncols) return;
    P3Region *rg = regs + (size_t)col * R;
    const int nw = (M + 63) / 64;
    int openCarry = 0, byteCarry = 0;                        // position 0 opens the first run
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        const P3Region v = (r < R) ? rg[r] : P3Region{-1, -1, 0, 0};
        const int inc = wave_iscan_max(v.lastT + 1);
        const int prevLast = lane_shr1(inc, 0) - 1;          // last transition in the earlier regions of this group of 64, -1 = none
        const int open = (prevLast >= 0) ? prevLast : openCarry;
        const bool ownsLast = (r < R) && (r * words_per_region < nw) && ((r + 1) * words_per_region >= nw);
        const int wb = (r < R) ? (v.firstT >= 0 ? p3_nbytes(v.firstT - open) : 0) + v.inner + (ownsLast ? p3_nbytes(M - (v.lastT >= 0 ? v.lastT : open)) : 0) : 0;
        const int incB = wave_iscan_sum(wb);
        if (r < R) rg[r] = P3Region{open, byteCarry + incB - wb, 0, 0};
        const int lastAll = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (lastAll >= 0) openCarry = lastAll;
        byteCarry += __builtin_amdgcn_readlane(incB, 63);
    }
    if (lane == 0) colBytes[col] = (unsigned long long)byteCarry;
}

template <int IT>
__global__ __launch_bounds__(BLOCK) void p3r_emit_kernel(const unsigned long long *ycols, int wpc64, int M, int R, const P3Region *regs,
                                                           const unsigned long long *colOff, uint8_t *out) {
    const int col = blockIdx.y, reg = blockIdx.x * WAVES + wave_id(), lane = lane_id();
    if (reg >= R) return;
    const unsigned long long *y = ycols + (size_t)col * wpc64;
    const int nw = (M + 63) / 64, base = reg * 64 * IT;
    if (base >= nw) return;
    unsigned long long cur[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { const int wd = base + i * 64 + lane; cur[i] = (wd < nw) ? y[wd] : 0ULL; }
    const P3Region rg = regs[(size_t)col * R + reg];
    uint8_t *obase = out + colOff[col] + rg.lastT;           // .lastT holds the region's byte base after the combine
    int prevHi = (base > 0) ? (int)(y[base - 1] >> 32) : 0;
    int carryT = rg.firstT;                                  // .firstT holds the start of the run open at the region's start
    int done = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int wd = base + i * 64 + lane;
        const int hiPrevIter = prevHi;
        unsigned long long tr = p3r_transitions<IT>(cur, i, base, lane, M, prevHi);
        const int tl = tr ? wd * 64 + 63 - __clzll(tr) : -1;
        const int inc = wave_iscan_max(tl + 1);
        int st = max(lane_shr1(inc, 0) - 1, carryT);
        const bool lastWord = (wd == nw - 1);
        int bytes = 0;
        { int s2 = st; for (unsigned long long t2 = tr; t2; t2 &= t2 - 1) { const int pz = wd * 64 + __ffsll((long long)t2) - 1; bytes += p3_nbytes(pz - s2); s2 = pz; } if (lastWord) bytes += p3_nbytes(M - s2); }
        const int incB = wave_iscan_sum(bytes);
        const int ph = lane_shr1((int)(cur[i] >> 32), hiPrevIter);   // cross-lane: outside the divergent branch below
        if (bytes) {
            uint8_t *o = obase + done + incB - bytes;
            unsigned v = (wd == 0) ? (unsigned)(cur[i] & 1ULL) : ((unsigned)ph >> 31);   // value of the run open at this word = the last bit before it
            for (; tr; tr &= tr - 1) { const int pz = wd * 64 + __ffsll((long long)tr) - 1; o = p3_emit(o, v, pz - st); st = pz; v ^= 1u; }
            if (lastWord) p3_emit(o, v, M - st);
        }
        done += __builtin_amdgcn_readlane(incB, 63);
        const int wl = __builtin_amdgcn_readlane(inc, 63) - 1;
        if (wl >= 0) carryT = wl;
    }
}

// ---------------------------------------------------------------------------------------------
// pack3 decode (unpack3, pbwtCore.c:279-305).
__device__ __forceinline__ int p3_len(uint8_t b) {
    b &= 0x7f;
    return b < 64 ? b : (b < 96 ? (b - 64) << 6 : (b - 96) << 11);
}
// pass 1: per block of DEC_CHUNK bytes, total run length
constexpr int DEC_CHUNK = 4096;
__global__ __launch_bounds__(BLOCK) void dec_sum_kernel(const uint8_t *z, size_t nz, unsigned long long *blockSum) {
    __shared__ unsigned long long s_w[WAVES];
    const size_t b0 = (size_t)blockIdx.x * DEC_CHUNK;
    unsigned long long s = 0;
    for (int q = threadIdx.x; q < DEC_CHUNK; q += BLOCK) { const size_t i = b0 + q; if (i < nz) s += (unsigned)p3_len(z[i]); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane_id() == 0) s_w[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) { s = 0; for (int q = 0; q < WAVES; ++q) s += s_w[q]; blockSum[blockIdx.x] = s; }
}
// pass 2: with exclusive block offsets: colStart[c] = byte index of the first byte of column c
// (start position divisible by M and non-empty run); colStart[N] = nz written by the host.
__global__ __launch_bounds__(BLOCK) void dec_colstart_kernel(const uint8_t *z, size_t nz, const unsigned long long *blockOff,
                                                            int M, long long N, long long *colStart) {
    __shared__ unsigned long long s_w[WAVES];
    __shared__ unsigned long long s_carry;
    const size_t b0 = (size_t)blockIdx.x * DEC_CHUNK;
    if (threadIdx.x == 0) s_carry = blockOff[blockIdx.x];
    __syncthreads();
    for (int q0 = 0; q0 < DEC_CHUNK; q0 += BLOCK) {
        const size_t i = b0 + q0 + threadIdx.x;
        const unsigned len = (i < nz) ? (unsigned)p3_len(z[i]) : 0u;
        unsigned long long inc = len;
        for (int o = 1; o < 64; o <<= 1) { unsigned long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
        const unsigned long long start = pre + inc - len;
        if (i < nz && len && start % (unsigned long long)M == 0) {
            const unsigned long long c = start / (unsigned long long)M;
            if ((long long)c < N) colStart[c] = (long long)i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}
// pass 2b: a well-formed panel has every column start found (a run never straddles a column boundary,
// pbwtCore.c:254-267), strictly increasing, at most M bytes per column.  Checked BEFORE any expand: a crafted file
// otherwise leaves colStart[c] = -1 (the memset) and the expand would index z[] and y[] out of bounds.
__global__ void dec_validate_kernel(const long long *colStart, long long N, long long nz, int M, int *err) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const long long bs = colStart[c], be = (c + 1 < N) ? colStart[c + 1] : nz;
    if (bs < 0 || be <= bs || be > nz || be - bs > (long long)M) atomicExch(err, 2);
}

// pass 3: expand columns [c0, c0+nc) into sorted bit columns (zero-initialised by the caller).
// One block per column; runs of ones set bits.  Malformed input (already rejected by dec_validate_kernel on the
// upload path) cannot write outside the column: bounds are re-checked and the accumulators are 64-bit.
__global__ __launch_bounds__(BLOCK) void dec_expand_kernel(const uint8_t *z, const long long *colStart, long long c0, int M,
                                                          unsigned long long *ycols, int wpc64, int *err) {
    __shared__ long long s_w[WAVES];
    __shared__ long long s_carry;
    const long long c = c0 + blockIdx.x;
    const long long bs = colStart[c], be = colStart[c + 1];
    unsigned long long *y = ycols + (size_t)blockIdx.x * wpc64;
    if (bs < 0 || be < bs || be - bs > (long long)M) { if (threadIdx.x == 0) atomicExch(err, 2); return; }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long b = bs; b < be; b += BLOCK) {
        const long long i = b + threadIdx.x;
        const uint8_t byte = (i < be) ? z[i] : 0;
        const int len = (i < be) ? p3_len(byte) : 0;
        long long inc = len;
        for (int o = 1; o < 64; o <<= 1) { long long v = __shfl_up(inc, o); if (lane_id() >= o) inc += v; }
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        long long pre = s_carry, tot = 0;
        for (int q = 0; q < WAVES; ++q) { if (q < wave_id()) pre += s_w[q]; tot += s_w[q]; }
        const long long start = pre + inc - len;
        if (start + len > (long long)M) atomicExch(err, 2);
        if (len && (byte & 0x80) && start < (long long)M) {
            int lo = (int)start, hi = (int)min(start + len, (long long)M);      // [lo,hi)
            while (lo < hi) {
                const int wd = lo >> 6, bo = lo & 63;
                const int take = min(64 - bo, hi - lo);
                const unsigned long long mk = (take == 64) ? ~0ULL : (((1ULL << take) - 1ULL) << bo);
                if (take == 64) y[wd] = mk; else atomicOr(&y[wd], mk);
                lo += take;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_carry != (long long)M) atomicExch(err, 3);
}

}  // namespace pbwtk
