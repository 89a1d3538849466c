/* pbwtGpu.c — the reference-side binding of libpbwtgpu.so (include/pbwt_amd.h): the translation unit a
 * richarddurbin/pbwt maintainer adds to route the whole-panel loops of the hot path to the MI355X engine.
 *
 * It is compiled against the reference's OWN pbwt.h and replaces the bodies of
 *     matchMaximalWithin          pbwtMatch.c:115-142
 *     matchLongWithin2            pbwtMatch.c:85-113   (static there: reached through pbwtLongMatches)
 *     matchSequencesSweep         pbwtMatch.c:363-443
 *     matchSequencesSweepSparse   pbwtMatch.c:501-602
 * and adds two entry points the per-column API cannot express:
 *     pbwtBuildFromBitColumns     the per-site loop of pbwtReadMacs, pbwtIO.c:477-483
 *     pbwtCursorAtSite            a PbwtCursor (pbwt.h:74-87) filled from the device sweep at site k
 * Everything else of the reference — pbwtLongMatches, reportMatch, -check, matchSequencesDynamic, the cursor
 * functions, file formats — stays the reference's code and calls into these.
 *
 * Two ways to build it (integration/Makefile does both, tests/test_integration.py checks both):
 *   (1) stand-alone object:  gcc -c -I<reference> -I<repo>/include pbwtGpu.c   — for a tree whose pbwtMatch.c has had
 *       the four bodies removed (or weakened at link time);
 *   (2) unity build, no edit of the reference at all: pbwtMatchGpu.c renames the four functions while it includes
 *       pbwtMatch.c, then includes this file with PBWT_GPU_UNITY defined — which also gives the replacement access to
 *       pbwtMatch.c's file-static `matchLengthHist` (pbwtMatch.c:28), i.e. the -stats histogram of pbwtLongMatches.
 *
 * Error behaviour is the reference's: every failure of the library becomes die() (utils.c:31-44).  There is no CPU
 * fall-back here or in the library. */
#ifndef PBWT_GPU_UNITY
#include "pbwt.h"		/* no include guard in the reference: the unity build already has it through pbwtMatch.c */
#endif
#include "pbwt_amd.h"

/* after pbwtMatchGpu.c's renaming these are not declared under their real names in this TU */
void matchMaximalWithin (PBWT *p, void (*report)(int ai, int bi, int start, int end)) ;
void matchSequencesSweep (PBWT *p, PBWT *q, void (*report)(int ai, int bi, int start, int end)) ;
void matchSequencesSweepSparse (PBWT *p, PBWT *q, int nSparse,
				void (*report)(int ai, int bi, int start, int end, BOOL isSparse)) ;

static pbwtamd_engine *gpuEngine = 0 ;
static int gpuEngineM = 0 ;

static pbwtamd_engine *engineFor (int M)	/* one cached engine per panel width: the reference is single-threaded */
{
  if (gpuEngine && gpuEngineM != M) { pbwtamd_engine_destroy (gpuEngine) ; gpuEngine = 0 ; }
  if (!gpuEngine && pbwtamd_engine_create (&gpuEngine, 0, M, 0, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  gpuEngineM = M ;
  return gpuEngine ;
}

void pbwtGpuRelease (void)	/* optional: free the device state before exit */
{ if (gpuEngine) { pbwtamd_engine_destroy (gpuEngine) ; gpuEngine = 0 ; gpuEngineM = 0 ; } }

#define PZ(p) arrp((p)->yz,0,uchar), (int64_t) arrayMax((p)->yz)

/* replaces pbwtMatch.c:115-142 */
void matchMaximalWithin (PBWT *p, void (*report)(int ai, int bi, int start, int end))
{
  if (!p || !p->yz) die ("matchMaximalWithin called without a PBWT") ;
#ifdef PBWT_GPU_UNITY
  if (matchLengthHist)		/* the -stats branch, pbwtMatch.c:130-131 */
    { int64_t *h = mycalloc (p->N+2, int64_t) ; int i ;
      if (pbwtamd_max_within (engineFor(p->M), PZ(p), p->N, p->aFstart, 0, 0, 0, h, p->N+2))
	die ("pbwt_amd: %s", pbwtamd_last_error()) ;
      for (i = 0 ; i <= p->N ; ++i)
	if (h[i]) array(matchLengthHist, i, int) += (int) h[i] ;
      free (h) ;
      return ;
    }
#endif
  if (pbwtamd_max_within (engineFor(p->M), PZ(p), p->N, p->aFstart, report, 0, 0, 0, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
}

/* replaces pbwtMatch.c:85-113; static like the original when it lives in pbwtMatch.c's TU */
#ifdef PBWT_GPU_UNITY
static
#endif
void matchLongWithin2 (PBWT *p, int T, void (*report)(int ai, int bi, int start, int end))
{
  if (pbwtamd_long_within (engineFor(p->M), PZ(p), p->N, p->aFstart, T, report, 0, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
}

/* the "no match to query" lines the reference writes from inside its sweep (pbwtMatch.c:405-410, 489-494), in its order */
static void noMatchLog (PBWT *p, int64_t nomatch)
{
  int32_t *ev = 0 ; int64_t nev = 0, i ;
  if (!nomatch) return ;
  if (pbwtamd_get_nomatch_events (engineFor(p->M), &ev, &nev)) die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  for (i = 0 ; i < nev ; ++i)
    fprintf (logFile, "no match to query %d value %d at site %d\n", ev[4*i], ev[4*i+1], ev[4*i+2]) ;
  if (nev < nomatch)		/* the library keeps a bounded number of events; the count is exact */
    fprintf (logFile, "... %lld further no-match events not listed\n", (long long) (nomatch - nev)) ;
  pbwtamd_free (ev) ;
}

static void sweepLog (PBWT *q, int64_t *tot)	/* pbwtMatch.c:438-439 */
{
  fprintf (logFile, "Average number of best matches including alternates %.1f, Average length %.1f, Av number per position %.1f\n",
	   tot[0]/(double)q->M, tot[1]/(double)tot[0], tot[1]/(double)(q->M*q->N)) ;
}

/* replaces pbwtMatch.c:363-443 */
void matchSequencesSweep (PBWT *p, PBWT *q, void (*report)(int ai, int bi, int start, int end))
{
  int64_t nomatch, tot[2] ;
  if (q->N != p->N) die ("query length in matchSequences %d != PBWT length %d", q->N, p->N) ;
  if (pbwtamd_match_sweep (engineFor(p->M), PZ(p), p->N, p->aFstart,
			   q->M, PZ(q), q->aFstart, report, 0, 0, &nomatch, tot))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  noMatchLog (p, nomatch) ;
  sweepLog (q, tot) ;
}

/* replaces pbwtMatch.c:501-602 (BOOL is a char in utils.h: the library's 5th callback argument is an int) */
static void (*sparseReport)(int ai, int bi, int start, int end, BOOL isSparse) ;
static void sparseThunk (int ai, int bi, int start, int end, int isSparse)
{ (*sparseReport) (ai, bi, start, end, (BOOL) isSparse) ; }

void matchSequencesSweepSparse (PBWT *p, PBWT *q, int nSparse,
				void (*report)(int ai, int bi, int start, int end, BOOL isSparse))
{
  int64_t nomatch, tot[2] ;
  if (q->N != p->N) die ("query length in matchSequences %d != PBWT length %d", q->N, p->N) ;
  sparseReport = report ;
  if (pbwtamd_match_sweep_sparse (engineFor(p->M), PZ(p), p->N, p->aFstart,
				  q->M, PZ(q), q->aFstart, nSparse, sparseThunk, 0, 0, &nomatch, tot))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  noMatchLog (p, nomatch) ;
  sweepLog (q, tot) ;
}

/* the per-site loop of pbwtReadMacs (pbwtIO.c:477-483: y[j] = x[a[j]] ; pbwtCursorWriteForwards) for N sites at
   once: cols = the parsed x[] of each site packed as bit h of column k (wpc 32-bit words per column).
   Appends to an empty PBWT made by pbwtCreate (M, 0); sets p->yz, p->aFend, p->N like the loop does. */
void pbwtBuildFromBitColumns (PBWT *p, const uint32_t *cols, int wpc, int N)
{
  uint8_t *yz ; int64_t nz ;
  if (p->N) die ("pbwtBuildFromBitColumns: the PBWT already holds %d sites", p->N) ;
  if (!p->aFend) p->aFend = myalloc (p->M, int) ;
  if (pbwtamd_build (engineFor(p->M), cols, wpc, N, 0, p->aFstart, &yz, &nz, p->aFend, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  if (p->yz) arrayDestroy (p->yz) ;
  p->yz = arrayCreate (nz+1, uchar) ;
  if (nz) { array(p->yz, nz-1, uchar) = 0 ; memcpy (arrp(p->yz, 0, uchar), yz, nz) ; }	/* array() sets arrayMax */
  pbwtamd_free (yz) ;
  p->N = N ;
}

/* A reference PbwtCursor positioned before site k by the device: the state pbwtCursorCreate (p, TRUE, TRUE) plus k
   calls of pbwtCursorForwardsReadAD leave (pbwtCore.c:420-445,543-557), so per-column code can carry on from there
   with pbwtCursorForwardsReadAD (u, k), pbwtCursorCalculateU, pbwtCursorMap ... on the CPU. */
PbwtCursor *pbwtCursorAtSite (PBWT *p, int k)
{
  PbwtCursor *u = pbwtCursorCreate (p, TRUE, TRUE) ;
  int64_t nBlockStart, n ;
  if (k < 0 || k > p->N) die ("pbwtCursorAtSite: site %d outside 0..%d", k, p->N) ;
  if (pbwtamd_cursor_at (engineFor(p->M), PZ(p), p->N, p->aFstart, k,
			 u->a, u->d, u->y, &u->c, u->u, &nBlockStart, &n))
    die ("pbwt_amd: %s", pbwtamd_last_error()) ;
  u->nBlockStart = nBlockStart ; u->n = n ; u->isBlockEnd = (k < p->N) ? TRUE : FALSE ;
  return u ;
}
