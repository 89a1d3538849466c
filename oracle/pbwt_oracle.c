/* pbwt_oracle.c — CPU restatement of the PBWT hot path.  TEST INFRASTRUCTURE ONLY.
 * See pbwt_oracle.h for the role and the parity status (PINNED against oracle/_ref).
 * Written from the algorithm descriptions in SURVEY.md §8(a) and the reference semantics;
 * every function cites the reference lines it restates.  Nothing here is shipped or called by
 * the product path. */
#include "pbwt_oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ pack3 codec */
/* Three-level run-length code, one byte = one run (pbwtCore.c:216-238):
 * top bit = value; low 7 bits b: b<64 -> n=b; 64<=b<96 -> n=(b-64)<<6; b>=96 -> n=(b-96)<<11. */
#define P3_MAX1 64
#define P3_MAX2 (32 << 6)
#define P3_MAX3 (31 << 11)

static inline int p3_len(uint8_t b)
{
    b &= 0x7f;
    if (b < 64) return b;
    if (b < 96) return (b - 64) << 6;
    return (b - 96) << 11;
}

/* emit one run (pbwtCore.c:240-252) */
static inline size_t p3_emit(uint8_t v, int n, uint8_t *out)
{
    uint8_t *o = out, top = (uint8_t)(v << 7);
    while (n >= P3_MAX3) { *o++ = top | 0x7f; n -= P3_MAX3; }
    if (n >= P3_MAX2) { *o++ = top | 0x60 | (uint8_t)(n >> 11); n &= 0x7ff; }
    if (n >= P3_MAX1) { *o++ = top | 0x40 | (uint8_t)(n >> 6); n &= 0x3f; }
    if (n) *o++ = top | (uint8_t)n;
    return (size_t)(o - out);
}

/* pack M values (pbwtCore.c:254-267); does not need the y[M] sentinel (bounds-checked) */
size_t orc_pack3(const uint8_t *y, int M, uint8_t *out)
{
    uint8_t *o = out;
    int m = 0;
    while (m < M) {
        int m0 = m;
        uint8_t v = y[m++];
        while (m < M && y[m] == v) ++m;
        o += p3_emit(v, m - m0, o);
    }
    return (size_t)(o - out);
}

/* unpack until M values are produced (pbwtCore.c:279-305) */
size_t orc_unpack3(const uint8_t *z, int M, uint8_t *y, int *n0)
{
    const uint8_t *zp = z;
    int m = 0, zeros = 0;
    while (m < M) {
        uint8_t b = *zp++;
        int n = p3_len(b);
        uint8_t v = b >> 7;
        if (m + n > M) n = M - m;           /* reference would overrun; never happens on valid data */
        memset(y + m, v, (size_t)n);
        if (!v) zeros += n;
        m += n;
    }
    if (n0) *n0 = zeros;
    return (size_t)(zp - z);
}

/* ------------------------------------------------------------------ cursor */
/* pbwtNakedCursorCreate (pbwtCore.c:402-418): a = aInit or identity, d = 0 except d[0]=d[M]=1 */
void orc_cursor_init(int M, const int32_t *aInit, int32_t *a, int32_t *d)
{
    for (int i = 0; i < M; ++i) a[i] = aInit ? aInit[i] : i;
    if (d) {
        memset(d, 0, sizeof(int32_t) * (size_t)(M + 1));
        d[0] = 1; d[M] = 1;
    }
}

/* pbwtCursorForwardsA (pbwtCore.c:458-470): stable 0/1 partition of a */
void orc_step_A(int M, const uint8_t *y, int32_t *a, int32_t *b)
{
    int u = 0, v = 0;
    for (int i = 0; i < M; ++i) {
        if (y[i] == 0) a[u++] = a[i];
        else           b[v++] = a[i];
    }
    memcpy(a + u, b, sizeof(int32_t) * (size_t)v);
}

/* pbwtCursorForwardsAD (pbwtCore.c:485-508): partition + running-max divergence update */
void orc_step_AD(int M, int k, const uint8_t *y, int32_t *a, int32_t *d, int32_t *b, int32_t *e)
{
    int u = 0, v = 0;
    int32_t p = k + 1, q = k + 1;
    for (int i = 0; i < M; ++i) {
        if (d[i] > p) p = d[i];
        if (d[i] > q) q = d[i];
        if (y[i] == 0) { a[u] = a[i]; d[u] = p; ++u; p = 0; }
        else           { b[v] = a[i]; e[v] = q; ++v; q = 0; }
    }
    memcpy(a + u, b, sizeof(int32_t) * (size_t)v);
    memcpy(d + u, e, sizeof(int32_t) * (size_t)v);
    d[0] = k + 2; d[M] = k + 2;
}

/* pbwtCursorCalculateU (pbwtCore.c:510-519) */
int orc_calc_u(int M, const uint8_t *y, int32_t *u)
{
    int c = 0;
    for (int i = 0; i < M; ++i) { u[i] = c; if (y[i] == 0) ++c; }
    u[M] = c;
    return c;
}

/* ------------------------------------------------------------------ checksums + generator */
static inline uint64_t sm64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

uint64_t orc_checksum_i32(const int32_t *v, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += sm64(((uint64_t)i << 32) | (uint32_t)v[i]);
    return s;
}

uint64_t orc_checksum_u8(const uint8_t *v, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += sm64(((uint64_t)i << 32) | (uint32_t)v[i]);
    return s;
}

static inline uint64_t h2(uint64_t seed, uint64_t a, uint64_t b)
{
    return sm64(sm64(seed ^ (a * 0xD1B54A32D192ED03ULL)) + b);
}

#define SYN_SEGLEN 2048u
#define SYN_MUT_THR 4294967u   /* ~1e-3 * 2^32 */

static inline uint32_t syn_site_thr(uint64_t seed, uint64_t k)
{
    uint64_t hk = h2(seed ^ 0xB, k, 0);
    uint32_t e = (uint32_t)(hk & 0xff) % 11u;
    uint32_t base = 1u << (31 - e);
    return base / 2 + (uint32_t)((hk >> 8) % (base / 2));
}

static inline uint64_t syn_founder_word(uint64_t seed, uint64_t k)
{
    uint32_t thr = syn_site_thr(seed, k);
    uint64_t w = 0;
    for (uint64_t f = 0; f < 64; ++f)
        if ((uint32_t)(h2(seed ^ 0xA, f, k) >> 32) < thr) w |= 1ULL << f;
    return w;
}

static inline uint32_t syn_allele(uint64_t seed, int kind, uint64_t fw, uint64_t h, uint64_t k)
{
    if (kind == 1) return (uint32_t)(h2(seed ^ 0xE, h, k) >> 63);
    uint64_t off = h2(seed ^ 0xD, h, 0) % SYN_SEGLEN;
    uint64_t seg = (k + off) / SYN_SEGLEN;
    uint32_t F = (uint32_t)(h2(seed ^ 0xC, h, seg) & 63);
    uint32_t mut = ((uint32_t)(h2(seed ^ 0xE, h, k) >> 32) < SYN_MUT_THR) ? 1u : 0u;
    return ((uint32_t)(fw >> F) & 1u) ^ mut;
}

void orc_synth_bitcols(int M, int k0, int ncols, int wpc, uint64_t seed, int kind, uint32_t *bits)
{
    for (int j = 0; j < ncols; ++j) {
        uint64_t k = (uint64_t)(k0 + j);
        uint64_t fw = (kind == 0) ? syn_founder_word(seed, k) : 0;
        uint32_t *col = bits + (size_t)j * (size_t)wpc;
        memset(col, 0, sizeof(uint32_t) * (size_t)wpc);
        for (int h = 0; h < M; ++h)
            if (syn_allele(seed, kind, fw, (uint64_t)h, k)) col[h >> 5] |= 1u << (h & 31);
    }
}

/* ------------------------------------------------------------------ build (pbwtIO.c:477-483) */
static int site_dump_index(const int32_t *sites, int n, int k)
{
    for (int i = 0; i < n; ++i) if (sites[i] == k) return i;
    return -1;
}

int orc_build_bitcols(int M, int ncols, int k0, const uint32_t *bits, int wpc, int with_d,
                      int32_t *a_io, int32_t *d_io,
                      uint8_t *yz, size_t yzcap, size_t *nz, int32_t *aFend,
                      uint64_t *csum_a, uint64_t *csum_d,
                      const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump)
{
    int32_t *a = malloc(sizeof(int32_t) * (size_t)M);
    int32_t *b = malloc(sizeof(int32_t) * (size_t)M);
    int32_t *d = malloc(sizeof(int32_t) * (size_t)(M + 1));
    int32_t *e = malloc(sizeof(int32_t) * (size_t)(M + 1));
    uint8_t *y = malloc((size_t)M + 1);
    size_t n = 0;
    int rc = 0;

    if (a_io) { memcpy(a, a_io, sizeof(int32_t) * (size_t)M); }
    else orc_cursor_init(M, NULL, a, NULL);
    if (d_io) memcpy(d, d_io, sizeof(int32_t) * (size_t)(M + 1));
    else { memset(d, 0, sizeof(int32_t) * (size_t)(M + 1)); d[0] = d[M] = k0 + 1; }
    y[M] = 2;                                   /* Y_SENTINEL (pbwt.h:143) */

    for (int j = 0; j <= ncols; ++j) {
        int k = k0 + j;
        if (csum_a) csum_a[j] = orc_checksum_i32(a, (size_t)M);
        if (csum_d && with_d) csum_d[j] = orc_checksum_i32(d, (size_t)M + 1);
        int di = site_dump_index(dump_sites, ndump, k);
        if (di >= 0) {
            if (a_dump) memcpy(a_dump + (size_t)di * (size_t)M, a, sizeof(int32_t) * (size_t)M);
            if (d_dump && with_d)
                memcpy(d_dump + (size_t)di * (size_t)(M + 1), d, sizeof(int32_t) * (size_t)(M + 1));
        }
        if (j == ncols) break;
        const uint32_t *col = bits + (size_t)j * (size_t)wpc;
        for (int i = 0; i < M; ++i) {           /* y[j] = x[a[j]]  (pbwtIO.c:478) */
            int32_t h = a[i];
            y[i] = (uint8_t)((col[h >> 5] >> (h & 31)) & 1u);
        }
        if (yz) {                               /* pack3arrayAdd (pbwtCore.c:269-277) */
            if (n + (size_t)M > yzcap) { rc = -1; break; }
            n += orc_pack3(y, M, yz + n);
        }
        if (with_d) orc_step_AD(M, k, y, a, d, b, e);   /* WriteForwardsAD (pbwtCore.c:580-585) */
        else        orc_step_A(M, y, a, b);              /* WriteForwards   (pbwtCore.c:573-578) */
    }
    if (nz) *nz = n;
    if (aFend) memcpy(aFend, a, sizeof(int32_t) * (size_t)M);   /* pbwtCursorToAFend (:587-591) */
    if (a_io) memcpy(a_io, a, sizeof(int32_t) * (size_t)M);
    if (d_io) memcpy(d_io, d, sizeof(int32_t) * (size_t)(M + 1));
    free(a); free(b); free(d); free(e); free(y);
    return rc;
}

/* ------------------------------------------------------------------ read-side cursor */
typedef struct {
    int M;
    const uint8_t *z; size_t nz, n;   /* packed columns, read offset */
    uint8_t *y; int c;
    int32_t *a, *d, *b, *e, *u;
} ocursor;

/* pbwtCursorCreate(p, TRUE, TRUE) (pbwtCore.c:420-445) */
static void oc_open(ocursor *x, int M, const uint8_t *z, size_t nz, const int32_t *aStart)
{
    x->M = M; x->z = z; x->nz = nz; x->n = 0; x->c = 0;
    x->y = malloc((size_t)M + 1); x->y[M] = 2; memset(x->y, 0, (size_t)M);
    x->a = malloc(sizeof(int32_t) * (size_t)M);
    x->b = malloc(sizeof(int32_t) * (size_t)M);
    x->d = malloc(sizeof(int32_t) * (size_t)(M + 1));
    x->e = malloc(sizeof(int32_t) * (size_t)(M + 1));
    x->u = malloc(sizeof(int32_t) * (size_t)(M + 1));
    orc_cursor_init(M, aStart, x->a, x->d);
    if (nz) x->n = orc_unpack3(z, M, x->y, &x->c);
}

static void oc_close(ocursor *x)
{
    free(x->y); free(x->a); free(x->b); free(x->d); free(x->e); free(x->u);
}

/* the "read next column unless at end" tail of ForwardsRead/ReadAD (pbwtCore.c:527-557);
 * in a forward-only sweep isBlockEnd is always TRUE here so only the second branch matters,
 * and y stays stale once the bytes are exhausted. */
static void oc_read_next(ocursor *x)
{
    if (x->n < x->nz) x->n += orc_unpack3(x->z + x->n, x->M, x->y, &x->c);
}

static void oc_forwards_read_AD(ocursor *x, int k)
{
    orc_step_AD(x->M, k, x->y, x->a, x->d, x->b, x->e);
    oc_read_next(x);
}

/* ForwardsRead uses the run-driven ForwardsAPacked (pbwtCore.c:595-619) whose result equals the
 * plain stable partition */
static void oc_forwards_read(ocursor *x)
{
    orc_step_A(x->M, x->y, x->a, x->b);
    oc_read_next(x);
}

int orc_sweep_AD(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                 uint64_t *csum_a, uint64_t *csum_d, uint64_t *csum_y,
                 const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump,
                 uint8_t *y_dump, int32_t *c_dump)
{
    ocursor u;
    oc_open(&u, M, yz, nz, aFstart);
    for (int k = 0; k <= N; ++k) {
        if (csum_a) csum_a[k] = orc_checksum_i32(u.a, (size_t)M);
        if (csum_d) csum_d[k] = orc_checksum_i32(u.d, (size_t)M + 1);
        if (csum_y) csum_y[k] = orc_checksum_u8(u.y, (size_t)M);
        int di = site_dump_index(dump_sites, ndump, k);
        if (di >= 0) {
            if (a_dump) memcpy(a_dump + (size_t)di * (size_t)M, u.a, sizeof(int32_t) * (size_t)M);
            if (d_dump) memcpy(d_dump + (size_t)di * (size_t)(M + 1), u.d, sizeof(int32_t) * (size_t)(M + 1));
            if (y_dump) memcpy(y_dump + (size_t)di * (size_t)M, u.y, (size_t)M);
            if (c_dump) c_dump[di] = u.c;
        }
        oc_forwards_read_AD(&u, k);
    }
    oc_close(&u);
    return 0;
}

/* ------------------------------------------------------------------ record buffer */
static void mv_push(orc_matchvec *mv, int ai, int bi, int start, int end)
{
    if (mv->n == mv->cap) {
        mv->cap = mv->cap ? mv->cap * 2 : 1024;
        mv->v = realloc(mv->v, mv->cap * sizeof(orc_match));
    }
    orc_match *r = &mv->v[mv->n++];
    r->ai = ai; r->bi = bi; r->start = start; r->end = end;
}

/* ------------------------------------------------------------------ matchMaximalWithin */
/* pbwtMatch.c:115-142.  The two scans walk away from i while the divergence stays within the
 * best-match block; finding an equal allele there (and not being at the final site) means the
 * match extends, so i is skipped. */
/* one site of the sweep (the body of the loop over i, pbwtMatch.c:120-137): `live` = k < N.  hist mode (the -stats sink,
 * pbwtMatch.c:46-49 counts by length) or record mode (the report callback's arguments in callback order). */
static inline int within_site(int M, int k, int live, const uint8_t *y, const int32_t *a, const int32_t *d,
                              int mode, orc_matchvec *out, int64_t *hist, int histlen)
{
    for (int i = 0; i < M; ++i) {
        int m = i - 1, n = i + 1, skip = 0;
        if (d[i] <= d[i + 1])
            while (d[m + 1] <= d[i]) { if (y[m--] == y[i] && live) { skip = 1; break; } }
        if (!skip && d[i] >= d[i + 1])
            while (d[n] <= d[i + 1]) { if (y[n++] == y[i] && live) { skip = 1; break; } }
        if (skip) continue;
        if (mode == 1) {
            int len = (d[i] < d[i + 1]) ? k - d[i] : k - d[i + 1];
            if (len < 0 || len >= histlen) return -2;
            ++hist[len];
        } else {
            for (int j = m + 1; j < i; ++j) mv_push(out, a[i], a[j], d[i], k);
            for (int j = i + 1; j < n; ++j) mv_push(out, a[i], a[j], d[i + 1], k);
        }
    }
    return 0;
}

static int max_within_range(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                            int mode, orc_matchvec *out, int64_t *hist, int histlen, int k_lo, int k_hi)
{
    ocursor u;
    int rc = 0;
    if (M < 2) return -3;
    oc_open(&u, M, yz, nz, aFstart);
    for (int k = 0; k <= N; ++k) {
        if (k >= k_lo && k < k_hi) {
            rc = within_site(M, k, k < N, u.y, u.a, u.d, mode, out, hist, histlen);
            if (rc) break;
        }
        oc_forwards_read_AD(&u, k);
    }
    oc_close(&u);
    return rc;
}

/* A BLOCK of sites of the build + -stats maxWithin job, continued from a checkpointed cursor (a_k0, d_k0): per site the
 * build loop's gather and pack3 (pbwtIO.c:477-483, pbwtCore.c:269-277), the sweep of matchMaximalWithin at that site
 * (pbwtMatch.c:120-137, histogram sink) and WriteForwardsAD (pbwtCore.c:580-585); when the block reaches the panel's end
 * (k0 + ncols == n_total) the closing sweep at k == N (pbwtMatch.c:118 runs k <= N; `live` is false there so the stale
 * column is never consulted).  Lets a full-length run be checked block by block from the device's own checkpoints: block i
 * starts from checkpoint i and must arrive at checkpoint i+1.  No global state: callable from several threads at once. */
int orc_segment(int M, int ncols, int k0, int n_total, const uint32_t *bits, int wpc,
                int32_t *a_io, int32_t *d_io, uint8_t *yz, size_t yzcap, size_t *nz, int64_t *hist, int histlen)
{
    int32_t *a = a_io, *d = d_io;
    int32_t *b = malloc(sizeof(int32_t) * (size_t)M);
    int32_t *e = malloc(sizeof(int32_t) * (size_t)(M + 1));
    uint8_t *y = malloc((size_t)M + 1);
    size_t n = 0;
    int rc = 0;
    if (M < 2 || d[0] != k0 + 1 || d[M] != k0 + 1) { rc = -3; goto done; }
    y[M] = 2;                                   /* Y_SENTINEL (pbwt.h:143) */
    for (int j = 0; j < ncols; ++j) {
        const int k = k0 + j;
        const uint32_t *col = bits + (size_t)j * (size_t)wpc;
        for (int i = 0; i < M; ++i) {
            int32_t h = a[i];
            y[i] = (uint8_t)((col[h >> 5] >> (h & 31)) & 1u);
        }
        if (hist && (rc = within_site(M, k, 1, y, a, d, 1, NULL, hist, histlen))) goto done;
        if (yz) {
            if (n + (size_t)M > yzcap) { rc = -1; goto done; }
            n += orc_pack3(y, M, yz + n);
        }
        orc_step_AD(M, k, y, a, d, b, e);
    }
    if (hist && k0 + ncols == n_total) rc = within_site(M, n_total, 0, y, a, d, 1, NULL, hist, histlen);
done:
    if (nz) *nz = n;
    free(b); free(e); free(y);
    return rc;
}

int orc_max_within(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                   int mode, orc_matchvec *out, int64_t *hist, int histlen)
{
    return max_within_range(M, N, yz, nz, aFstart, mode, out, hist, histlen, 0, N + 1);
}

/* the same sweep with the report callback filtered to the sites k_lo <= k < k_hi (what a caller's
 * report() does when it only wants the matches ending in a window): full-width parity checks at
 * M = 1 M without materialising every site's records */
int orc_max_within_range(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                         int k_lo, int k_hi, orc_matchvec *out)
{
    return max_within_range(M, N, yz, nz, aFstart, 0, out, NULL, 0, k_lo, k_hi);
}

/* ------------------------------------------------------------------ matchLongWithin2 */
/* pbwtMatch.c:85-113 (-longWithin L).  Positions are cut into blocks wherever d[i] > k-L; when a
 * block is closed and holds both alleles, every pair ia < ib in it with different alleles is
 * reported with start = max d over (ia, ib].  Faithful to two quirks of the reference: i0/na/nb live
 * across sites, so the block still open at the end of a site is never reported (at the next site
 * d[0] closes it with an empty pair loop); and at k == N the alleles are the stale column N-1. */
int orc_long_within(int M, int N, int L, const uint8_t *yz, size_t nz, const int32_t *aFstart, orc_matchvec *out)
{
    ocursor u;
    int i0 = 0, na = 0, nb = 0;
    oc_open(&u, M, yz, nz, aFstart);
    for (int k = 0; k <= N; ++k) {
        const int32_t *d = u.d, *a = u.a;
        const uint8_t *y = u.y;
        for (int i = 0; i < M; ++i) {
            if (d[i] > k - L) {
                if (na && nb)
                    for (int ia = i0; ia < i; ++ia) {
                        int dmin = 0;
                        for (int ib = ia + 1; ib < i; ++ib) {
                            if (d[ib] > dmin) dmin = d[ib];
                            if (y[ib] != y[ia]) mv_push(out, a[ia], a[ib], dmin, k);
                        }
                    }
                na = 0; nb = 0; i0 = i;
            }
            if (y[i] == 0) na++; else nb++;
        }
        oc_forwards_read_AD(&u, k);
    }
    oc_close(&u);
    return 0;
}

/* ------------------------------------------------------------------ matchSequencesSweep */
/* pbwtMatch.c:363-443 */
int orc_match_sweep(int Mp, int N, const uint8_t *pz, size_t pnz, const int32_t *pStart,
                    int Mq, const uint8_t *qz, size_t qnz, const int32_t *qStart,
                    orc_matchvec *out, int64_t *n_nomatch, int64_t *tot)
{
    ocursor up, uq;
    oc_open(&up, Mp, pz, pnz, pStart);
    oc_open(&uq, Mq, qz, qnz, qStart);
    int32_t *f = calloc((size_t)Mq, sizeof(int32_t));
    int32_t *dq = calloc((size_t)Mq, sizeof(int32_t));
    int64_t nTot = 0, totLen = 0, nomatch = 0;
    const int M = Mp;

    for (int k = 0; k < N; ++k) {
        const uint8_t *py = up.y;
        const int32_t *pd = up.d, *pa = up.a;
        for (int j = 0; j < Mq; ++j) {
            int jj = uq.a[j];
            uint8_t x = uq.y[j];
            if (py[f[jj]] == x) continue;
            /* an equally long match further down that does extend? (pbwtMatch.c:381-383) */
            int iPlus = f[jj], found = 0;
            while (++iPlus < M && pd[iPlus] <= dq[jj])
                if (py[iPlus] == x) { f[jj] = iPlus; found = 1; break; }
            if (found) continue;
            /* no: these matches end here (pbwtMatch.c:385-386) */
            for (int i = f[jj]; i < iPlus; ++i) mv_push(out, jj, pa[i], dq[jj], k);
            nTot += iPlus - f[jj]; totLen += (int64_t)(k - dq[jj]) * (iPlus - f[jj]);
            /* widen [iMinus,iPlus] by the smaller divergence until an x is met (:389-411) */
            int iMinus = f[jj];
            int dPlus = (iPlus < M) ? pd[iPlus] : k;
            int dMinus = pd[iMinus];
            for (;;) {
                if (dMinus <= dPlus) {
                    int hit = -1;
                    while (pd[iMinus] <= dMinus)          /* pd[0] = k+1 stops this */
                        if (py[--iMinus] == x) hit = iMinus;
                    if (hit >= 0) { f[jj] = hit; dq[jj] = dMinus; break; }
                    dMinus = pd[iMinus];
                } else {
                    int got = 0;
                    while (iPlus < M && pd[iPlus] <= dPlus) {
                        if (py[iPlus] == x) { f[jj] = iPlus; dq[jj] = dPlus; got = 1; break; }
                        ++iPlus;
                    }
                    if (got) break;
                    dPlus = (iPlus == M) ? k : pd[iPlus];
                    if (!iMinus && iPlus == M) { ++nomatch; dq[jj] = k + 1; break; }
                }
            }
        }
        /* LF-map every query's position (pbwtMatch.c:416-423) */
        up.c = orc_calc_u(M, up.y, up.u);
        for (int j = 0; j < Mq; ++j) {
            int jj = uq.a[j];
            int i = f[jj];
            f[jj] = uq.y[j] ? up.c + i - up.u[i] : up.u[i];     /* pbwtCursorMap (pbwt.h:130-131) */
            if (f[jj] == M) f[jj] = 0;
        }
        oc_forwards_read_AD(&up, k);
        oc_forwards_read(&uq);
    }
    /* matches running to the end (pbwtMatch.c:430-436) */
    for (int j = 0; j < Mq; ++j) {
        int jj = uq.a[j], i;
        mv_push(out, jj, up.a[f[jj]], dq[jj], N);
        for (i = f[jj]; ++i < M && up.d[i] <= dq[jj]; ) mv_push(out, jj, up.a[i], dq[jj], N);
        nTot += i - f[jj]; totLen += (int64_t)(N - dq[jj]) * (i - f[jj]);
    }
    if (n_nomatch) *n_nomatch = nomatch;
    if (tot) { tot[0] = nTot; tot[1] = totLen; }
    free(f); free(dq);
    oc_close(&up); oc_close(&uq);
    return 0;
}

/* ------------------------------------------------------------------ matchSequencesSweepSparse */
static void mv5_push(orc_match5vec *v, int ai, int bi, int start, int end, int sparse)
{
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->v = realloc(v->v, v->cap * sizeof(orc_match5)); }
    orc_match5 m = { ai, bi, start, end, sparse };
    v->v[v->n++] = m;
}

/* reportAndUpdate (pbwtMatch.c:452-499): pd/py/pa = the cursor's d, y, a; for a sparse cursor k runs at
 * 1/nS of the rate: reported starts are nS*d + k%nS, the open end uses k/nS */
static void orc_report_update(int jj, int k, uint8_t x, int M, const int32_t *pd, const uint8_t *py, const int32_t *pa,
                              int32_t *f, int32_t *dq, int isSparse, int nS, orc_match5vec *out,
                              int64_t *nTot, int64_t *totLen, int64_t *nomatch)
{
    const int kend = isSparse ? k / nS : k;
    int iPlus = f[jj];
    while (++iPlus < M && pd[iPlus] <= dq[jj])
        if (py[iPlus] == x) { f[jj] = iPlus; return; }
    const int dj = isSparse ? nS * dq[jj] + k % nS : dq[jj];
    for (int i = f[jj]; i < iPlus; ++i) mv5_push(out, jj, pa[i], dj, k, isSparse);
    *nTot += iPlus - f[jj]; *totLen += (int64_t)(k - dj) * (iPlus - f[jj]);
    int iMinus = f[jj];
    int dPlus = (iPlus < M) ? pd[iPlus] : kend;
    int dMinus = pd[iMinus];
    for (;;) {
        if (dMinus <= dPlus) {
            int hit = -1;
            while (pd[iMinus] <= dMinus)                  /* pd[0] = (k or k/nS)+1 stops this */
                if (py[--iMinus] == x) hit = iMinus;
            if (hit >= 0) { f[jj] = hit; dq[jj] = dMinus; return; }
            dMinus = pd[iMinus];
        } else {
            while (iPlus < M && pd[iPlus] <= dPlus) {
                if (py[iPlus] == x) { f[jj] = iPlus; dq[jj] = dPlus; return; }
                ++iPlus;
            }
            dPlus = (iPlus < M) ? pd[iPlus] : kend;
            if (!iMinus && iPlus == M) { ++*nomatch; dq[jj] = 1 + kend; return; }
        }
    }
}

int orc_match_sweep_sparse(int Mp, int N, const uint8_t *pz, size_t pnz, const int32_t *pStart,
                           int Mq, const uint8_t *qz, size_t qnz, const int32_t *qStart, int nSparse,
                           orc_match5vec *out, int64_t *n_nomatch, int64_t *tot)
{
    const int M = Mp, nS = nSparse > 1 ? nSparse : 0;
    ocursor up, uq;
    oc_open(&up, Mp, pz, pnz, pStart);
    oc_open(&uq, Mq, qz, qnz, qStart);
    int32_t *f = calloc((size_t)Mq, sizeof(int32_t)), *dq = calloc((size_t)Mq, sizeof(int32_t));
    ocursor *upp = NULL; int32_t **ff = NULL, **dd = NULL; uint8_t *xp = NULL;
    if (nS) {
        upp = calloc((size_t)nS, sizeof(ocursor)); ff = calloc((size_t)nS, sizeof(int32_t *)); dd = calloc((size_t)nS, sizeof(int32_t *));
        for (int kk = 0; kk < nS; ++kk) {                 /* pbwtNakedCursorCreate (M, 0) (pbwtCore.c:402-418) */
            oc_open(&upp[kk], M, NULL, 0, NULL);
            ff[kk] = calloc((size_t)Mq, sizeof(int32_t)); dd[kk] = calloc((size_t)Mq, sizeof(int32_t));
        }
        xp = malloc((size_t)M);
    }
    int64_t nTot = 0, totLen = 0, nomatch = 0;
    for (int k = 0; k < N; ++k) {
        const int kk = nS ? k % nS : 0;
        if (nS) {                                         /* the sparse cursor's column: this site's alleles in ITS order (:537-541) */
            for (int j = 0; j < M; ++j) xp[up.a[j]] = up.y[j];
            for (int j = 0; j < M; ++j) upp[kk].y[j] = xp[upp[kk].a[j]];
        }
        for (int j = 0; j < Mq; ++j) {
            const int jj = uq.a[j];
            const uint8_t xq = uq.y[j];
            if (up.y[f[jj]] != xq) orc_report_update(jj, k, xq, M, up.d, up.y, up.a, f, dq, 0, nS, out, &nTot, &totLen, &nomatch);
            if (nS && upp[kk].y[ff[kk][jj]] != xq)
                orc_report_update(jj, k, xq, M, upp[kk].d, upp[kk].y, upp[kk].a, ff[kk], dd[kk], 1, nS, out, &nTot, &totLen, &nomatch);
        }
        up.c = orc_calc_u(M, up.y, up.u);
        for (int j = 0; j < Mq; ++j) {
            const int jj = uq.a[j], i = f[jj];
            f[jj] = uq.y[j] ? up.c + i - up.u[i] : up.u[i];
            if (f[jj] == M) f[jj] = 0;
        }
        if (nS) {
            ocursor *us = &upp[kk];
            us->c = orc_calc_u(M, us->y, us->u);
            for (int j = 0; j < Mq; ++j) {
                const int jj = uq.a[j], i = ff[kk][jj];
                ff[kk][jj] = uq.y[j] ? us->c + i - us->u[i] : us->u[i];
                if (ff[kk][jj] == M) ff[kk][jj] = 0;
            }
            orc_step_AD(M, k / nS, us->y, us->a, us->d, us->b, us->e);
        }
        oc_forwards_read_AD(&up, k);
        oc_forwards_read(&uq);
    }
    for (int j = 0; j < Mq; ++j) {                        /* dense tails (:577-583) */
        int jj = uq.a[j], i;
        mv5_push(out, jj, up.a[f[jj]], dq[jj], N, 0);
        for (i = f[jj]; ++i < M && up.d[i] <= dq[jj]; ) mv5_push(out, jj, up.a[i], dq[jj], N, 0);
        nTot += i - f[jj]; totLen += (int64_t)(N - dq[jj]) * (i - f[jj]);
    }
    for (int kk = 0; kk < nS; ++kk)                       /* sparse tails (:585-594) */
        for (int j = 0; j < Mq; ++j) {
            int jj = uq.a[j], i;
            const int dj = nS * dd[kk][jj] + kk;
            mv5_push(out, jj, upp[kk].a[ff[kk][jj]], dj, N, 1);
            for (i = ff[kk][jj]; ++i < M && upp[kk].d[i] <= dd[kk][jj]; ) mv5_push(out, jj, upp[kk].a[i], dj, N, 1);
            nTot += i - ff[kk][jj]; totLen += (int64_t)(N - dd[kk][jj]) * (i - ff[kk][jj]);
        }
    if (n_nomatch) *n_nomatch = nomatch;
    if (tot) { tot[0] = nTot; tot[1] = totLen; }
    free(f); free(dq);
    for (int kk = 0; kk < nS; ++kk) { oc_close(&upp[kk]); free(ff[kk]); free(dd[kk]); }
    free(upp); free(ff); free(dd); free(xp);
    oc_close(&up); oc_close(&uq);
    return 0;
}

/* ------------------------------------------------------------------ -haps (pbwtIO.c:839-857) */
int orc_haplotypes(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart, uint8_t *out)
{
    ocursor u;
    oc_open(&u, M, yz, nz, aFstart);
    for (int k = 0; k < N; ++k) {
        uint8_t *row = out + (size_t)k * (size_t)M;
        for (int j = 0; j < M; ++j) row[u.a[j]] = u.y[j];
        oc_forwards_read(&u);
    }
    oc_close(&u);
    return 0;
}
