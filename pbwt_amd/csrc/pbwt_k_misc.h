// pbwt_k_misc.h — synthetic panels, per-site checksums, small conversions.
// Part of the kernel set of pbwt_kernels.h (include that, not this file: the parts build on each other in its order).
#pragma once

namespace pbwtk {

// ---------------------------------------------------------------------------------------------
// synthetic panel generator (SURVEY.md §8d recipe in integer arithmetic; the test checker restates it)
__device__ __forceinline__ uint64_t h2(uint64_t seed, uint64_t a, uint64_t b) {
    return sm64(sm64(seed ^ (a * 0xD1B54A32D192ED03ULL)) + b);
}

__global__ __launch_bounds__(BLOCK) void synth_kernel(uint32_t *bits, int M, int k0, int ncols, int wpc,
                                                     uint64_t seed, int kind) {
    __shared__ uint64_t s_fw;
    const int col = blockIdx.y;
    const uint64_t k = (uint64_t)(k0 + col);
    if (kind == 0) {
        // founder word for this site: bit f = founder f carries the derived allele
        if (threadIdx.x < 64) {
            const uint64_t hk = h2(seed ^ 0xB, k, 0);
            const uint32_t e = (uint32_t)(hk & 0xff) % 11u;
            const uint32_t bse = 1u << (31 - e);
            const uint32_t thr = bse / 2 + (uint32_t)((hk >> 8) % (bse / 2));
            const bool on = (uint32_t)(h2(seed ^ 0xA, (uint64_t)threadIdx.x, k) >> 32) < thr;
            const unsigned long long m = __ballot(on);
            if (threadIdx.x == 0) s_fw = m;
        }
        __syncthreads();
    }
    const uint64_t fw = (kind == 0) ? s_fw : 0;
    for (int wd = blockIdx.x * BLOCK + threadIdx.x; wd < wpc; wd += gridDim.x * BLOCK) {
        uint32_t out = 0;
        for (int b = 0; b < 32; ++b) {
            const uint64_t h = (uint64_t)wd * 32 + b;
            if (h >= (uint64_t)M) break;
            uint32_t al;
            if (kind == 1) al = (uint32_t)(h2(seed ^ 0xE, h, k) >> 63);
            else {
                const uint64_t off = h2(seed ^ 0xD, h, 0) % 2048u;
                const uint64_t seg = (k + off) / 2048u;
                const uint32_t F = (uint32_t)(h2(seed ^ 0xC, h, seg) & 63);
                const uint32_t mut = ((uint32_t)(h2(seed ^ 0xE, h, k) >> 32) < 4294967u) ? 1u : 0u;
                al = ((uint32_t)(fw >> F) & 1u) ^ mut;
            }
            out |= al << b;
        }
        bits[(size_t)col * wpc + wd] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// per-site checksums over ring slots: csum[site] += sum_i sm64(i<<32 | v[i]); grid (tiles, sites)
__global__ __launch_bounds__(BLOCK) void checksum_kernel(const int *A, const int *D, size_t strideA, size_t strideD,
                                                        int M, int with_d, unsigned long long *ca,
                                                        unsigned long long *cd, unsigned long long *cy, int y_valid_sites, int packed = 0) {
    __shared__ unsigned long long s_red[WAVES][3];
    const int site = blockIdx.y;
    const int *a = A + (size_t)site * strideA;
    const int *d = D + (size_t)site * strideD;
    unsigned long long sa = 0, sd = 0, sy = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i <= M; i += gridDim.x * BLOCK) {
        if (packed) {                                       // slots hold d | y << 31 and no ids (PBWTAMD_PACKED_CHECKSUM: the packed fill checked position by position)
            const int v = d[i];
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
