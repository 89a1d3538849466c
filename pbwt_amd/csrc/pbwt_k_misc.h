// pbwt_k_misc.h — synthetic panels, per-site checksums, small conversions.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// synthetic panel generator (SURVEY.md §8d recipe in integer arithmetic; the test checker restates it)
__device__ __forceinline__ uint64_t h2(uint64_t seed, uint64_t a, uint64_t b) {
    return sm64(sm64(seed ^ (a * 0xD1B54A32D192ED03ULL)) + b);
}

// A thread owns ONE haplotype and walks SYNTH_CPB consecutive columns: what depends on the haplotype alone (its offset into the 2048-site
// segments) or changes once per segment (its founder) stays in registers, so a (haplotype, site) cell costs one hash — the mutation draw — instead
// of three; the allele bits of 64 haplotypes leave as one ballot.  Same integer arithmetic cell by cell as the oracle's restatement
// (the test-side restatement of the same generator lives with the checker).  grid (ceil(M / 256), ceil(ncols / SYNTH_CPB)).
constexpr int SYNTH_CPB = 32;
__global__ __launch_bounds__(BLOCK) void synth_kernel(uint32_t *bits, int M, int k0, int ncols, int wpc,
                                                     uint64_t seed, int kind) {
    __shared__ uint64_t s_fw[SYNTH_CPB];
    const int t = threadIdx.x, lane = lane_id(), c0 = blockIdx.y * SYNTH_CPB, nc = min(SYNTH_CPB, ncols - c0);
    if (kind == 0) {
        // founder word of every site of the block: bit f = founder f carries the derived allele
        if (t < 64) {
            for (int c = 0; c < nc; ++c) {
                const uint64_t k = (uint64_t)(k0 + c0 + c);
                const uint64_t hk = h2(seed ^ 0xB, k, 0);
                const uint32_t e = (uint32_t)(hk & 0xff) % 11u;
                const uint32_t bse = 1u << (31 - e);
                const uint32_t thr = bse / 2 + (uint32_t)((hk >> 8) % (bse / 2));
                const bool on = (uint32_t)(h2(seed ^ 0xA, (uint64_t)t, k) >> 32) < thr;
                const unsigned long long m = __ballot(on);
                if (t == 0) s_fw[c] = m;
            }
        }
        __syncthreads();
    }
    const uint64_t h = (uint64_t)blockIdx.x * BLOCK + t;
    const bool valid = h < (uint64_t)M;
    const int wd = (int)((h - lane) >> 5);                  // first of the two 32-bit words the wave's 64 haplotypes fill
    const uint64_t off = (kind == 0) ? h2(seed ^ 0xD, h, 0) % 2048u : 0;
    uint64_t curseg = ~0ULL; uint32_t F = 0;
    for (int c = 0; c < nc; ++c) {
        const uint64_t k = (uint64_t)(k0 + c0 + c);
        uint32_t al;
        if (kind == 1) al = (uint32_t)(h2(seed ^ 0xE, h, k) >> 63);
        else {
            const uint64_t seg = (k + off) / 2048u;
            if (seg != curseg) { F = (uint32_t)(h2(seed ^ 0xC, h, seg) & 63); curseg = seg; }
            const uint32_t mut = ((uint32_t)(h2(seed ^ 0xE, h, k) >> 32) < 4294967u) ? 1u : 0u;
            al = ((uint32_t)(s_fw[c] >> F) & 1u) ^ mut;
        }
        const unsigned long long m = __ballot(valid && al);
        if (lane == 0) {
            uint32_t *row = bits + (size_t)(c0 + c) * wpc;
            if (wd < wpc) row[wd] = (uint32_t)m;
            if (wd + 1 < wpc) row[wd + 1] = (uint32_t)(m >> 32);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-site checksums over ring slots: csum[site] += sum_i sm64(i<<32 | v[i]); grid (tiles, sites)
__global__ __launch_bounds__(BLOCK) void checksum_kernel(const int *A, const int *D, size_t strideA, size_t strideD,
                                                        int M, int with_d, unsigned long long *ca,
                                                        unsigned long long *cd, unsigned long long *cy, int y_valid_sites, int packed = 0,
                                                        const unsigned short *P16 = nullptr, size_t stride16 = 0, int kbase = 0) {
    __shared__ unsigned long long s_red[WAVES][3];
    const int site = blockIdx.y;
    const int *a = A + (size_t)site * strideA;
    const int *d = D + (size_t)site * strideD;
    unsigned long long sa = 0, sd = 0, sy = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i <= M; i += gridDim.x * BLOCK) {
        if (packed) {                                       // slots hold d | y << 31 and no ids (PBWTAMD_PACKED_CHECKSUM: the packed fill checked position by position)
            const int v = (packed == 2) ? p16_word(P16[(size_t)site * stride16 + i], kbase + site + 1, d, i) : d[i];     // (2: the 16-bit ring, escapes from d)
            if (i < M) sy += sm64(((uint64_t)i << 32) | ((site < y_valid_sites) ? ((uint32_t)v >> 31) : 0u));
            sd += sm64(((uint64_t)i << 32) | (uint32_t)(i < M ? (v & 0x7fffffff) : v));
            continue;
        }
        if (i < M) {
            const int v = a[i];
            sa += sm64(((uint64_t)i << 32) | (uint32_t)(v & AMASK));
            const uint32_t y = (site < y_valid_sites) ? ((uint32_t)v >> 31) : 0u;
            sy += sm64(((uint64_t)i << 32) | y);
        }
        if (with_d) sd += sm64(((uint64_t)i << 32) | (uint32_t)d[i]);
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sd += __shfl_xor(sd, o); sy += __shfl_xor(sy, o); }
    if (lane_id() == 0) { s_red[wave_id()][0] = sa; s_red[wave_id()][1] = sd; s_red[wave_id()][2] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sa = sd = sy = 0;
        for (int q = 0; q < WAVES; ++q) { sa += s_red[q][0]; sd += s_red[q][1]; sy += s_red[q][2]; }
        atomicAdd(ca + site, sa);
        if (with_d) atomicAdd(cd + site, sd);
        atomicAdd(cy + site, sy);
    }
}

// alleles back to original haplotype order (pbwtWriteHaplotypes, pbwtIO.c:845: hap[a[j]] = y[j]);
// grid (tiles, sites); out[site][hap] = 0/1
__global__ __launch_bounds__(BLOCK) void unsort_alleles_kernel(const int *A, size_t strideA, int M, unsigned char *out) {
    const int s = blockIdx.y;
    const int *a = A + (size_t)s * strideA;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < M; j += gridDim.x * BLOCK) {
        const int v = a[j];
        out[(size_t)s * M + (v & AMASK)] = (unsigned char)((unsigned)v >> 31);
    }
}
__global__ void tags_to_bytes_kernel(const int *a, unsigned char *out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = (unsigned char)((unsigned)a[i] >> 31);
}

// strip tags: out[i] = a[i] & AMASK
__global__ void untag_kernel(const int *a, int *out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = a[i] & AMASK;
}

}  // namespace pbwtk
