/* pbwt_oracle.h — CPU restatement of the PBWT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for pbwt_amd: a plain-C, single-threaded restatement of the
 * reference algorithms (richarddurbin/pbwt) on the hot path.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (pbwt_amd/csrc -> libpbwtgpu.so) never links or calls anything here.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py and the
 * committed fixtures under tests/golden/) against outputs of the reference itself, compiled in
 * place from /root/reference into oracle/_ref/ by oracle/Makefile.
 *
 * Each function cites the reference file:line whose semantics it follows.
 */
#ifndef PBWT_ORACLE_H
#define PBWT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t ai, bi, start, end; } orc_match;

/* growable record buffer handed back to the caller (free with orc_free) */
typedef struct { orc_match *v; size_t n, cap; } orc_matchvec;

void orc_free(void *p);

/* ---- pack3 codec (pbwtCore.c:216-305) ---- */
size_t orc_pack3(const uint8_t *y, int M, uint8_t *out);            /* out needs >= M bytes */
size_t orc_unpack3(const uint8_t *z, int M, uint8_t *y, int *n0);    /* returns bytes consumed */

/* ---- cursor (pbwtCore.c:402-418, 458-470, 485-508, 510-519) ---- */
void orc_cursor_init(int M, const int32_t *aInit, int32_t *a, int32_t *d /* M+1 */);
void orc_step_A(int M, const uint8_t *y, int32_t *a, int32_t *b /* scratch M */);
void orc_step_AD(int M, int k, const uint8_t *y, int32_t *a, int32_t *d,
                 int32_t *b /* scratch M */, int32_t *e /* scratch M+1 */);
int  orc_calc_u(int M, const uint8_t *y, int32_t *u /* M+1 */);      /* returns c */

/* order-sensitive 64-bit checksum used for per-site parity at full size
 * (sum over i of splitmix64(i<<32 | (uint32)v[i]) mod 2^64; shared definition with the device) */
uint64_t orc_checksum_i32(const int32_t *v, size_t n);
uint64_t orc_checksum_u8(const uint8_t *v, size_t n);

/* ---- synthetic panel generator (same integer recipe as the device kernel) ----
 * bits: N columns of wpc 32-bit words, bit h of column k = allele of haplotype h at site k
 * (original haplotype order).  kind 0 = founder mosaic, kind 1 = iid Bernoulli(1/2). */
void orc_synth_bitcols(int M, int k0, int ncols, int wpc, uint64_t seed, int kind, uint32_t *bits);

/* ---- build from columns in original haplotype order: the pbwtReadMacs loop
 * (pbwtIO.c:477-483) with WriteForwards (A only, with_d=0) or WriteForwardsAD (with_d=1).
 * Outputs: yz (caller buffer, capacity yzcap, returns bytes in *nz), aFend[M], optionally
 * per-site checksums csum_a[N+1], csum_d[N+1] (state BEFORE step k, k=0..N), optionally full
 * dumps of a/d at the sites listed in dump_sites (ndump entries; a_dump ndump*M, d_dump
 * ndump*(M+1)).  a_io/d_io: optional initial and final cursor state (NULL = fresh cursor).
 * k0 = site index of the first column (for the d sentinels).  Returns 0, or -1 if yz overflowed. */
int orc_build_bitcols(int M, int ncols, int k0, const uint32_t *bits, int wpc, int with_d,
                      int32_t *a_io, int32_t *d_io,
                      uint8_t *yz, size_t yzcap, size_t *nz, int32_t *aFend,
                      uint64_t *csum_a, uint64_t *csum_d,
                      const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump);

/* ---- read-side sweep over a packed panel: cursor create + ForwardsReadAD loop
 * (pbwtCore.c:420-445, 543-557) exactly as matchMaximalWithin drives it (k=0..N inclusive).
 * Emits per-site checksums of a, d, y (arrays of N+1; y is stale at k=N like the reference) and
 * optional dumps. */
int orc_sweep_AD(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                 uint64_t *csum_a, uint64_t *csum_d, uint64_t *csum_y,
                 const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump,
                 uint8_t *y_dump, int32_t *c_dump);

/* ---- matchMaximalWithin (pbwtMatch.c:115-142).  mode 0: append every report() call to *out
 * (zero-length ones included, in callback order); mode 1: histogram (pbwtMatch.c:130-131) into
 * hist[0..histlen) (counts beyond histlen-1 are an error -> returns -2). */
int orc_max_within(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                   int mode, orc_matchvec *out, int64_t *hist, int histlen);

/* the same, reports filtered to the sites k_lo <= k < k_hi (mode 0 only) */
int orc_max_within_range(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart,
                         int k_lo, int k_hi, orc_matchvec *out);
/* a block of sites of build + -stats maxWithin continued from a checkpointed cursor (a_io/d_io in: state at k0, out: state at
 * k0+ncols); hist accumulates the block's reports (and the k == N sweep when the block ends the panel) */
int orc_segment(int M, int ncols, int k0, int n_total, const uint32_t *bits, int wpc,
                int32_t *a_io, int32_t *d_io, uint8_t *yz, size_t yzcap, size_t *nz, int64_t *hist, int histlen);

/* ---- matchLongWithin2 (pbwtMatch.c:85-113), the -longWithin L command: records in callback order */
int orc_long_within(int M, int N, int L, const uint8_t *yz, size_t nz, const int32_t *aFstart, orc_matchvec *out);

/* ---- matchSequencesSweep (pbwtMatch.c:363-443): panel p vs query panel q, both packed.
 * Appends report() calls in callback order; *n_nomatch counts the "no match to query" log
 * events (pbwtMatch.c:405-410); tot[0]=nTot, tot[1]=totLen (pbwtMatch.c:386,435). */
int orc_match_sweep(int Mp, int N, const uint8_t *pz, size_t pnz, const int32_t *pStart,
                    int Mq, const uint8_t *qz, size_t qnz, const int32_t *qStart,
                    orc_matchvec *out, int64_t *n_nomatch, int64_t *tot);

/* ---- matchSequencesSweepSparse (pbwtMatch.c:452-602): the dense sweep plus nSparse cursors over the
 * sites k = kk (mod nSparse), stepped with pbwtCursorForwardsAD(upp[kk], k/nSparse).  Records carry the
 * isSparse flag of the 5-argument callback; callback order kept (per site and query: dense block, then
 * sparse block; tails: dense for every query, then each sparse cursor in turn).  nSparse <= 1: dense only. */
typedef struct { int32_t ai, bi, start, end, sparse; } orc_match5;
typedef struct { orc_match5 *v; size_t n, cap; } orc_match5vec;
int orc_match_sweep_sparse(int Mp, int N, const uint8_t *pz, size_t pnz, const int32_t *pStart,
                           int Mq, const uint8_t *qz, size_t qnz, const int32_t *qStart, int nSparse,
                           orc_match5vec *out, int64_t *n_nomatch, int64_t *tot);

/* ---- haplotype recovery (-haps, pbwtIO.c:839-857): out[k*M + h] = allele (0/1) ---- */
int orc_haplotypes(int M, int N, const uint8_t *yz, size_t nz, const int32_t *aFstart, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
