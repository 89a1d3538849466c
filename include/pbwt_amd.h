/* pbwt_amd.h — C ABI of libpbwtgpu.so, the MI355X (gfx950) engine for the PBWT hot path.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point replaces a specific
 * whole-panel loop of the reference (richarddurbin/pbwt); the reference file:line each one stands
 * in for is cited on the declaration.  INTEGRATION.md shows the binding a reference maintainer
 * would add (a pbwtGpu.c that implements matchMaximalWithin() etc. on top of these calls).
 *
 * Conventions
 *   - return value 0 = success; nonzero = failure, message in pbwtamd_last_error() (the host
 *     binding turns that into the reference's die(), utils.c:31).  There is NO CPU fallback: if
 *     no gfx950 device is usable every compute entry point fails.
 *   - haplotype indices, site indices, a[] and d[] values are int32 like the reference's `int`
 *     (pbwt.h:36-37,81-82); byte offsets into packed columns are int64 like its `long`.
 *   - a "bit column" is one site's alleles, bit h of the column (little-endian 32-bit words,
 *     bit h&31 of word h>>5) = allele of haplotype/position h; columns are `wpc` words apart,
 *     wpc >= ceil(M/32).  "original order" columns are indexed by haplotype (x[] in
 *     pbwtIO.c:478), "sorted order" columns by PBWT position (u->y[], pbwt.h:78).
 *   - packed columns (`yz`) are the reference's pack3 run-length bytes (pbwtCore.c:216-225),
 *     byte-identical to PBWT.yz / the payload of a .pbwt file (pbwtIO.c:33-57).
 */
#ifndef PBWT_AMD_H
#define PBWT_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBWTAMD_ABI_VERSION 4   /* 2: pbwtamd_shard_* , pbwtamd_pass_advance_many, pbwtamd_get_nomatch_events (round 3-4); 3: pbwtamd_match_sweep_stream, pbwtamd_drain_packed (round 5); 4: pbwtamd_shard_stats (round 6; nothing removed or re-typed) */

typedef struct pbwtamd_engine pbwtamd_engine;

/* one report() invocation: (ai, bi, start, end) of pbwtMatch.c:46 / pbwt.h:208 */
typedef struct { int32_t ai, bi, start, end; } pbwtamd_match;

/* callback type of matchMaximalWithin / matchSequencesSweep (pbwt.h:208,213) */
typedef void (*pbwtamd_report_fn)(int ai, int bi, int start, int end);

/* ---- library / device ---- */
int         pbwtamd_abi_version(void);
const char *pbwtamd_last_error(void);
int         pbwtamd_device_count(void);            /* usable HIP devices (0 if none) */

/* ---- engine: device state for one panel of M haplotypes (the role of a PbwtCursor,
 * pbwt.h:74-87, plus its scratch, held in HBM).  batch_sites = sites processed per device
 * batch (0 = default).  stream = a hipStream_t to enqueue on, or NULL for the engine's own.
 * PROCESS-WIDE SIDE EFFECT: the first engine a process creates narrows the CALLING thread's CPU affinity — and with it that of every thread created while an engine
 * is alive, the HIP runtime's helpers included, which is the point — to one hardware thread per physical core of the GPU's NUMA node (the launching thread's
 * placement decides whether the chain runs 10 % slower as a whole).  The mask of that thread is restored when the process's last engine is
 * destroyed; threads the host created meanwhile keep what they inherited.  PBWTAMD_PIN=0 in the environment switches it off (a host with its own placement policy,
 * OpenMP teams, I/O threads: set it, and pin the launching thread yourself as pbwt_amd/pin.py does).  The card is found through HIP_VISIBLE_DEVICES /
 * ROCR_VISIBLE_DEVICES when they are index lists; anything else (UUIDs): no pinning. */
int  pbwtamd_engine_create(pbwtamd_engine **out, int device, int M, int batch_sites, void *stream);
void pbwtamd_engine_destroy(pbwtamd_engine *e);
int  pbwtamd_engine_M(const pbwtamd_engine *e);
int  pbwtamd_engine_wpc(const pbwtamd_engine *e);      /* words per bit column the engine uses */
int  pbwtamd_engine_batch(const pbwtamd_engine *e);

/* ======================================================================================
 * Host-buffer entry points: what the reference-side binding calls.  All are synchronous.
 * ====================================================================================== */

/* Build a PBWT from N bit columns in ORIGINAL haplotype order: the per-site loop of
 * pbwtReadMacs (pbwtIO.c:477-483; same shape at pbwtIO.c:573-574) = gather y[j]=x[a[j]],
 * pack3arrayAdd (pbwtCore.c:269-277) and pbwtCursorForwardsA (:458-470), or with with_d != 0
 * pbwtCursorForwardsAD (:485-508).
 *   aFstart  : initial order (PBWT.aFstart), NULL = identity (pbwtCreate, pbwtCore.c:46)
 *   yz_out   : receives a malloc()ed buffer with the packed columns (free with pbwtamd_free)
 *   aFend    : out, M ints = final a[] (pbwtCursorToAFend, pbwtCore.c:587-591)
 *   dFend    : out or NULL, M+1 ints = final d[] (only meaningful with_d) */
int pbwtamd_build(pbwtamd_engine *e, const uint32_t *bitcols, int wpc, int N, int with_d,
                  const int32_t *aFstart, uint8_t **yz_out, int64_t *nz_out,
                  int32_t *aFend, int32_t *dFend);

/* Forward sweep of a packed panel with divergence: pbwtCursorCreate(p,TRUE,TRUE) followed by
 * pbwtCursorForwardsReadAD(u,k) for k=0..N (pbwtCore.c:420-445,543-557), i.e. what every
 * read-side consumer drives.  Optional outputs (NULL to skip):
 *   csum_a/csum_d/csum_y [N+1] : order-sensitive checksums of a (M), d (M+1), y (M) at each k
 *                                (sum_i splitmix64(i<<32 | v[i]); y is all-zero at k=N here,
 *                                the reference leaves it stale)
 *   dump_sites[ndump]          : sites whose full a (ndump*M), d (ndump*(M+1)) and y (ndump*M
 *                                bytes, NULL to skip) are copied out — what exportSiteInfo
 *                                (pbwtMain.c:82-100) prints */
int pbwtamd_sweep_AD(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                     uint64_t *csum_a, uint64_t *csum_d, uint64_t *csum_y,
                     const int32_t *dump_sites, int ndump, int32_t *a_dump, int32_t *d_dump, uint8_t *y_dump);

/* The PbwtCursor struct (pbwt.h:74-87) at site k of a packed panel: fills the caller's cursor arrays with exactly what
 * pbwtCursorCreate(p,TRUE,TRUE) + k calls of pbwtCursorForwardsReadAD (pbwtCore.c:420-445,543-557) leave there —
 *   a[M], d[M+1] (sentinels d[0] = d[M] = k+1), y[M] (the column of site k; at k == N the stale column N-1, pbwtCore.c:539),
 *   *c = zeros in y, u[M+1] as pbwtCursorCalculateU (pbwtCore.c:510-519) computes it,
 *   *nBlockStart / *n = the cursor's byte offsets into yz (isBlockEnd = k < N) —
 * so a host caller can stop the device sweep at any site and go on with the per-column API (pbwt.h:114-127) on the CPU.
 * Any output pointer may be NULL. */
int pbwtamd_cursor_at(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, int k,
                      int32_t *a, int32_t *d, uint8_t *y, int32_t *c, int32_t *u, int64_t *nBlockStart, int64_t *n);

/* -haps: pbwtWriteHaplotypes (pbwtIO.c:839-857) without the text formatting: out[k*M + h] = allele
 * (0/1) of haplotype h at site k, recovered by a forward sweep of the packed panel */
int pbwtamd_haplotypes(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, uint8_t *out);

/* the same, streamed: sink(k0, nsites, rows, ctx) is called on the calling thread with the alleles of sites k0 .. k0+nsites-1
 * (nsites rows of M bytes), batch after batch in site order — pbwtWriteHaplotypes without an N x M matrix on the host */
int pbwtamd_haplotypes_stream(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                              void (*sink)(int k0, int nsites, const uint8_t *rows, void *ctx), void *ctx);

/* matchMaximalWithin (pbwtMatch.c:115-142): all set-maximal matches within the panel.
 * Exactly one of the three sinks is used:
 *   report   : called synchronously on the calling thread, in the reference's order (k, then i,
 *              then j; zero-length matches included — pbwtMatch.c:48 filters them itself)
 *   recs_out : receives a malloc()ed array of all reports in that order (+ count)
 *   hist     : the -stats histogram of pbwtMatch.c:130-131, hist[len] += 1 (histlen >= N+1) */
int pbwtamd_max_within(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                       pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out,
                       int64_t *hist, int histlen);

/* the same sweep, reports restricted to the sites k_lo <= end < k_hi (k_hi <= N+1; end == N is the closing
 * all-positions report): for callers whose report() only keeps a window of `end` values — the chain still runs
 * over every site, the record traffic is the window's */
int pbwtamd_max_within_range(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                             int k_lo, int k_hi, pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out);

/* matchLongWithin2 (pbwtMatch.c:85-113), the -longWithin L command: every pair of haplotypes
 * whose match ending at a site is at least L sites long, reported when the block closes; report
 * order and quirks are the reference's (see sweep_long_kernel).  L >= 1 (L == 0 is maxWithin). */
int pbwtamd_long_within(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart, int L,
                        pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out);

/* matchSequencesSweep (pbwtMatch.c:363-443): query panel q (Mq haplotypes, packed qz) against
 * the engine's panel (packed pz), both N sites.  Reports in the reference's order (k, query
 * PBWT order, i); n_nomatch counts the "no match to query" events (pbwtMatch.c:405-410);
 * tot[0]=nTot, tot[1]=totLen of pbwtMatch.c:386,435 (for the log line :438-439). */
int pbwtamd_match_sweep(pbwtamd_engine *e, const uint8_t *pz, int64_t pnz, int N, const int32_t *pStart,
                        int Mq, const uint8_t *qz, int64_t qnz, const int32_t *qStart,
                        pbwtamd_report_fn report, pbwtamd_match **recs_out, int64_t *nrecs_out,
                        int64_t *n_nomatch, int64_t *tot);

/* the "no match to query %d value %d at site %d" events (pbwtMatch.c:405-410, 489-494) of the last pbwtamd_match_sweep /
 * pbwtamd_match_sweep_sparse call on this engine, in the order the reference logs them: *n quadruples
 * (query jj, allele x, site k, isSparse) in a malloc()ed array (at most 65536 are kept; n_nomatch is exact) */
int pbwtamd_get_nomatch_events(pbwtamd_engine *e, int32_t **events, int64_t *n);

/* matchSequencesSweepSparse (pbwtMatch.c:501-602; declared pbwt.h:214): the query sweep against the
 * panel AND against nSparse sparse panels (the sites = kk mod nSparse, stepped with
 * pbwtCursorForwardsAD(upp[kk], k/nSparse)); reports carry the isSparse flag of the reference's
 * 5-argument callback and come in its order (per site and query: dense matches, then sparse; tails:
 * dense for every query, then each sparse cursor in turn).  nSparse <= 1: the dense sweep only. */
typedef struct { int32_t ai, bi, start, end, sparse; } pbwtamd_match5;
typedef void (*pbwtamd_report5_fn)(int ai, int bi, int start, int end, int isSparse);
int pbwtamd_match_sweep_sparse(pbwtamd_engine *e, const uint8_t *pz, int64_t pnz, int N, const int32_t *pStart,
                               int Mq, const uint8_t *qz, int64_t qnz, const int32_t *qStart, int nSparse,
                               pbwtamd_report5_fn report, pbwtamd_match5 **recs_out, int64_t *nrecs_out,
                               int64_t *n_nomatch, int64_t *tot);

/* matchSequencesSweep (pbwtMatch.c:363-443) STREAMED: neither panel is ever materialised — the form configs[4] (10^6 haplotypes x 10^7 sites, 10^4
 * queries) needs, where the packed panels alone would be O(100 GB) of host memory.  The library asks `cols` for the bit columns of both panels a batch at a
 * time and hands every batch's records to `recs`:
 *   cols(user, site0, ncols, &d_panel, &d_queries) -> 0: `ncols` consecutive columns of each panel from site0 on, DEVICE pointers on the engine's device,
 *        original haplotype order (the layout of pbwtamd_pass_advance: rows of pbwtamd_engine_wpc(e) words for the panel, of the same formula for Mq
 *        haplotypes — ((Mq + 31) / 32 rounded up to 4 — for the queries); contents complete when the callback returns; they must stay valid until the SECOND call after this one (two buffers, used alternately: the library prepares batch b + 1 beside batch b's sweep);
 *   recs(user, records, n) -> 0: the next n records (ai = query, bi = panel haplotype, start, end, sparse = 0) in the reference's callback order
 *        (k ascending, then the query panel's PBWT order, then i ascending; the tails at N last); valid during the call.
 * pStart / qStart: start orders (NULL = identity).  panel_opts — PBWTAMD_OPT_WITHIN_HIST | PBWTAMD_OPT_PACK3 | PBWTAMD_OPT_CHECKSUM — makes the SAME pass
 * over the panel feed those consumers too (histogram: pbwtamd_get_hist; bytes: pbwtamd_drain_packed, e.g. from inside `cols`).  n_nomatch, tot_out as
 * pbwtamd_match_sweep; the log lines' events through pbwtamd_get_nomatch_events. */
typedef int (*pbwtamd_cols_fn)(void *user, int site0, int ncols, const void **d_panel_cols, const void **d_query_cols);
typedef int (*pbwtamd_recs_fn)(void *user, const pbwtamd_match5 *records, int64_t n);
int pbwtamd_match_sweep_stream(pbwtamd_engine *e, int N, const int32_t *pStart, int Mq, const int32_t *qStart,
                               pbwtamd_cols_fn cols, pbwtamd_recs_fn recs, void *user, unsigned panel_opts,
                               int64_t *n_nomatch, int64_t *tot_out);

/* Query sharding across GPUs for matchSequencesSweep / -matchDynamic (queries are independent given the panel state:
 * pbwtMatch.c:376-414 touches f[jj], d[jj] of one query at a time).  After this call the query sweeps of `e` process the
 * queries lo <= jj < hi only (original indices in the query panel).  Every record's `sparse` field (and the isSparse column
 * of pbwtamd_get_nomatch_events) then carries, above bit 0, the query's rank in the query panel's PBWT order at the
 * record's site: (end, rank, isSparse) is the reference's emission order (k ascending, then uq->a order, dense before
 * sparse), so the streams of several ranks merge into exactly the reference's stream.  nTot / totLen / n_nomatch are this
 * range's share.  lo < 0: all queries again, plain `sparse` field.  Host side: pbwt_amd/queryshard.py. */
int pbwtamd_set_query_range(pbwtamd_engine *e, int lo, int hi);

/* Panel transforms — the "x[a[j]] = y[j]; y'[j] = x[a'[j]]; pbwtCursorWriteForwards" loops of pbwtBuildReverse
 * (pbwtCore.c:151-191), pbwtSubSample (pbwtSample.c:59-93), pbwtSubRange (pbwtCore.c:111-148), pbwtSelectSites /
 * pbwtRemoveSites (pbwtCore.c:623-732) — as one device pass: decode, regather, rebuild.  The new panel has
 *   n_out sites : output site j = input site site_order[j]   (NULL = all N sites in order)
 *   M_out haps  : output haplotype h = input haplotype hap_select[h]   (NULL = all M in order)
 *   start order aStart_out (NULL = identity);  outputs: packed columns (malloc()ed), final order aFend_out[M_out],
 *   and optionally the INPUT panel's final order aFend_fwd[M] (what pbwtBuildReverse starts the reverse cursor from).
 * BuildReverse: site_order = N-1..0, aStart_out = aFend_fwd of a first call (or the panel's stored aFend). */
int pbwtamd_regather(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, const int32_t *aFstart,
                     const int32_t *site_order, int n_out, const int32_t *hap_select, int M_out, const int32_t *aStart_out,
                     uint8_t **yz_out, int64_t *nz_out, int32_t *aFend_out, int32_t *aFend_fwd);

/* pack3 codec on the device (pbwtCore.c:254-305): N columns <-> packed bytes.
 * unpack returns sorted-order bit columns (N*wpc words, caller buffer). */
int pbwtamd_pack3(pbwtamd_engine *e, const uint32_t *sorted_bitcols, int wpc, int N,
                  uint8_t **yz_out, int64_t *nz_out);
int pbwtamd_unpack3(pbwtamd_engine *e, const uint8_t *yz, int64_t nz, int N, uint32_t *sorted_bitcols, int wpc);

void pbwtamd_free(void *p);

/* ======================================================================================
 * Device-buffer entry points (inputs already resident in HBM; used by bench.py and by callers
 * that keep panels on the device).  Pointers are device pointers on the engine's device; work is
 * enqueued on the engine's stream and is complete when pbwtamd_sync() returns.
 * ====================================================================================== */

/* synthetic panel (SURVEY.md §8d recipe): fills ncols bit columns for sites k0..k0+ncols-1.
 * kind 0 = founder mosaic, 1 = iid Bernoulli(1/2). */
int pbwtamd_synth_device(pbwtamd_engine *e, void *d_bitcols, int k0, int ncols, uint64_t seed, int kind);

/* start a pass at site k0: cursor state = aInit (host pointer, NULL=identity), d = 0 with the
 * sentinels d[0]=d[M]=k0+1 (pbwtCore.c:402-418) */
int pbwtamd_pass_begin(pbwtamd_engine *e, const int32_t *aInit, int k0, int n_total_sites);

#define PBWTAMD_OPT_WITH_D      1u   /* ForwardsAD instead of ForwardsA */
#define PBWTAMD_OPT_SORTED      2u   /* columns are in sorted (PBWT) order, not original order */
#define PBWTAMD_OPT_WITHIN_HIST 4u   /* fuse the maxWithin sweep, histogram sink */
#define PBWTAMD_OPT_CHECKSUM    8u   /* per-site checksums of a/d/y */
#define PBWTAMD_OPT_PACK3      16u   /* emit packed columns into the engine's yz buffer */
#define PBWTAMD_OPT_WITHIN_RECS 32u  /* fuse the maxWithin sweep, record sink */
#define PBWTAMD_OPT_LONG_RECS   64u  /* -longWithin L consumer (host entry point pbwtamd_long_within only) */

/* advance the pass over `ncols` more columns held at d_bitcols (device).  The pass must see the
 * column after the last one too unless it is the panel's last site (ncols_avail >= ncols+1); with
 * ncols_avail >= ncols+2 (or everything up to the panel's end) original-order passes run two sites
 * per launch.  Asynchronous. */
int pbwtamd_pass_advance(pbwtamd_engine *e, const void *d_bitcols, int wpc, int ncols, int ncols_avail,
                         unsigned opts);

/* Many panels per launch — the chromosomes of one cohort side by side.  P engines of the same width and batch size, all created on the
 * SAME stream (pbwtamd_engine_create's `stream` argument) and at the same site of their passes, advance over `ncols` columns each
 * (d_bitcols[p] = panel p's columns, same layout as pbwtamd_pass_advance) with every launch of the chain covering all P panels: below
 * ~250 000 haplotypes the per-site loop of pbwtReadMacs (pbwtIO.c:477-483) is bound by the cost of a dependent launch, which the panels
 * then share.  pass_begin / pass_end / the result getters stay per engine.  Falls back to one pbwtamd_pass_advance per engine where
 * the fused form does not apply (panels of more than 139 264 haplotypes, batches that are not a multiple of 8 sites, sorted columns). */
int pbwtamd_pass_advance_many(pbwtamd_engine **engines, int P, const void *const *d_bitcols, int wpc, int ncols, int ncols_avail, unsigned opts);

/* finish: runs the k==N sweep if a WITHIN sink is active; synchronises */
int pbwtamd_pass_end(pbwtamd_engine *e, unsigned opts);

/* close a pass before the panel's last site (no k == N sweep): for a caller that owns a block of sites only */
int pbwtamd_pass_stop(pbwtamd_engine *e);

/* right after pbwtamd_pass_begin(e, a_k, k0, N): also install the divergences d_k[0..M] of a checkpointed cursor
 * (d[0] = d[M] = k0+1, pbwtCore.c:507), so that a pass restarts at site k0 exactly where another one stopped */
int pbwtamd_pass_set_d(pbwtamd_engine *e, const int32_t *d);

int pbwtamd_sync(pbwtamd_engine *e);

/* results of the pass so far (host copies) */
int pbwtamd_get_state(pbwtamd_engine *e, int32_t *a /*M*/, int32_t *d /*M+1 or NULL*/);
int pbwtamd_get_hist(pbwtamd_engine *e, int64_t *hist, int histlen);
int pbwtamd_get_checksums(pbwtamd_engine *e, int k_first, int n, uint64_t *csum_a, uint64_t *csum_d, uint64_t *csum_y);
/* the pack3 bytes (PBWTAMD_OPT_PACK3) written since pass_begin = the p->yz array of pbwtCore.c:254-267; malloc'd, free with
 * pbwtamd_free */
int pbwtamd_get_packed(pbwtamd_engine *e, uint8_t **yz_out, int64_t *nz_out);
/* the same bytes handed over a piece at a time (what pbwtWrite fwrites, pbwtIO.c:33-57, for panels whose packed columns outgrow HBM): copies the bytes written
 * since pass_begin or the previous drain into buf (cap bytes) and rewinds the engine's buffer; *n = their number; buf == NULL: size only.  Synchronises the
 * consumer stream. */
int pbwtamd_drain_packed(pbwtamd_engine *e, uint8_t *buf, int64_t cap, int64_t *n);

/* timing of the chain kernel (the dominant kernel) over the last pass_advance calls since
 * pass_begin, measured with HIP events on the engine's stream: total ms and launches */
int pbwtamd_get_chain_timing(pbwtamd_engine *e, double *ms_total, int64_t *launches);
int pbwtamd_get_chain_sites(pbwtamd_engine *e, int64_t *sites);   /* sites those launches advanced (2 per launch on the build path) */

/* ======================================================================================
 * Position sharding of ONE panel across the GPUs of a node (BASELINE configs[3]): the recurrence of
 * pbwtCursorForwardsAD (pbwtCore.c:485-508) itself is split — rank g owns a contiguous range of positions of the sorted
 * order a_k, d_k.  Per round of 8 sites the ranks exchange one row of 256 (count, carry) pairs each (the exclusive scan of
 * the local 0/1 counts, for 8 sites at once) and every element of the new order is stored straight into its owner's memory
 * (the all-to-all, as peer stores over xGMI through hipIpc mappings).  The consumers (maxWithin sweep, pack3, checksums) are
 * sharded by site inside every batch.  One process per GPU:
 *     pbwtamd_engine_create(&e, device, M, batch, stream)          same M and batch on every rank
 *     pbwtamd_shard_init(e, rank, world, handles)                   -> PBWTAMD_SHARD_HANDLE_BYTES bytes to all-gather
 *     pbwtamd_shard_connect(e, all_handles)                         world blobs in rank order (after the all-gather)
 *     pbwtamd_pass_begin / pbwtamd_pass_advance / pbwtamd_pass_end  as on one GPU, with the SAME columns on every rank,
 *                                                                   original-order columns only
 * Afterwards every rank holds the complete final state (pbwtamd_get_state); the histogram and the per-site checksums of a
 * rank cover the sites it consumed — sum them over the ranks; the pack3 bytes of a rank are the blocks of sites
 * pbwtamd_shard_blocks lists, which concatenate in site order into PBWT.yz.  Several ranks may share one device (tests). */
#define PBWTAMD_SHARD_HANDLE_BYTES 576
int pbwtamd_shard_init(pbwtamd_engine *e, int rank, int world, void *handles_out);

/* What this rank's chain has spent waiting for its peers since the engine was created (position sharding; SURVEY 8e(1), the exchange step of pbwtCore.c:485-508 split
 * over ranks): out = {10 ns ticks waiting for the peers' rows of a round, number of such waits, ticks inside flag barriers, number of barriers}.  Synchronises the
 * engine's chain stream. */
int pbwtamd_shard_stats(pbwtamd_engine *e, uint64_t out[4]);
int pbwtamd_shard_connect(pbwtamd_engine *e, const void *all_handles);
/* positions [*pos_lo, *pos_hi) of the sorted order that `rank` owns */
int pbwtamd_shard_range(const pbwtamd_engine *e, int rank, int *pos_lo, int *pos_hi);
/* the blocks of sites whose packed columns this rank wrote since pass_begin, in order: first site, sites, and the offset in
 * pbwtamd_get_packed's buffer at which the block ENDS.  *n = number of blocks (call with cap = 0 to size the arrays). */
int pbwtamd_shard_blocks(pbwtamd_engine *e, int64_t *site0, int64_t *nsites, int64_t *byte_end, int cap, int *n);

/* diagnostics: with PBWTAMD_PROFILE=1 in the environment at engine creation the step kernel
 * stamps wall_clock64() (100 MHz) at its phase boundaries for every tile of the LAST launch:
 * out[tile*8 + phase], phases 0..6.  Returns the number of tiles copied, or < 0 on error. */
int pbwtamd_get_phase_profile(pbwtamd_engine *e, int64_t *out, int ntiles);

#ifdef __cplusplus
}
#endif
#endif
