/* gpu_driver.c — test-only entry points of oracle/_ref/libpbwtref_gpu.so: the reference compiled from its own
 * sources with integration/pbwtMatchGpu.c in place of pbwtMatch.c, so that the reference's OWN callers
 * (pbwtLongMatches, matchSequencesDynamic, reportMatch, -check) run with the MI355X engine underneath.
 * oracle/ref_driver.c (linked into the same library) already exposes pbwtLongMatches / matchSequencesSweep /
 * matchSequencesSweepSparse; this file adds the callers it lacks.  Our own code; TEST INFRASTRUCTURE. */
#include "pbwt.h"
#include <stdint.h>
#include <unistd.h>

void ref_init (void) ;						/* ref_driver.c */
void pbwtBuildFromBitColumns (PBWT *p, const uint32_t *cols, int wpc, int N) ;	/* pbwtGpu.c */
PbwtCursor *pbwtCursorAtSite (PBWT *p, int k) ;

int refgpu_is_gpu_build (void) { return 1 ; }

/* `pbwt -read panel.pbwt -matchDynamic query.pbwt > out` (pbwtMain.c:412-413 -> pbwtMatch.c:352-357) */
int refgpu_match_dynamic_to_file (const char *panel_pbwt, const char *query_pbwt, const char *out)
{
  ref_init () ;
  FILE *fp = fopen (panel_pbwt, "r") ; if (!fp) return -1 ;
  PBWT *p = pbwtRead (fp) ; fclose (fp) ;
  FILE *fq = fopen (query_pbwt, "r") ; if (!fq) return -2 ;
  fflush (stdout) ;
  int saved = dup (1) ;
  FILE *f = fopen (out, "w") ; if (!f) return -3 ;
  dup2 (fileno (f), 1) ;
  matchSequencesDynamic (p, fq) ;
  fflush (stdout) ;
  dup2 (saved, 1) ; close (saved) ; fclose (f) ; fclose (fq) ;
  pbwtDestroy (p) ;
  return 0 ;
}

/* the pbwtReadMacs site loop on the device, then what pbwtWrite would store */
long refgpu_build_bitcols (int M, int N, const uint32_t *bits, int wpc, uint8_t *yz_out, long yzcap, int32_t *aFend)
{
  ref_init () ;
  PBWT *p = pbwtCreate (M, 0) ;
  pbwtBuildFromBitColumns (p, bits, wpc, N) ;
  long nz = arrayMax (p->yz) ;
  if (nz > yzcap || p->N != N) nz = -1 ;
  else { memcpy (yz_out, arrp (p->yz, 0, uchar), nz) ; memcpy (aFend, p->aFend, sizeof (int) * M) ; }
  pbwtDestroy (p) ;
  return nz ;
}

/* a reference PbwtCursor positioned at site k by the device, then stepped `nsteps` further by the reference's own
   pbwtCursorForwardsReadAD on the CPU; returns the struct's fields at k (u_k, pos_k = {n, nBlockStart, isBlockEnd, c})
   and the state after the CPU steps */
int refgpu_cursor_continue (int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart, int k, int nsteps,
			    int32_t *a_k, int32_t *d_k, uint8_t *y_k, int32_t *u_k, long *pos_k,
			    int32_t *a_end, int32_t *d_end, uint8_t *y_end, int32_t *c_end)
{
  ref_init () ;
  PBWT *p = pbwtCreate (M, N) ;
  if (aFstart) memcpy (p->aFstart, aFstart, sizeof (int) * M) ;
  p->yz = arrayCreate (nz + 1, uchar) ;
  if (nz) memcpy (arrp (p->yz, 0, uchar), yz, nz) ;
  arrayMax (p->yz) = nz ;
  PbwtCursor *u = pbwtCursorAtSite (p, k) ;
  memcpy (a_k, u->a, sizeof (int) * M) ; memcpy (d_k, u->d, sizeof (int) * (M + 1)) ;
  memcpy (y_k, u->y, M) ; memcpy (u_k, u->u, sizeof (int) * (M + 1)) ;
  pos_k[0] = u->n ; pos_k[1] = u->nBlockStart ; pos_k[2] = u->isBlockEnd ; pos_k[3] = u->c ;
  if (u->y[M] != 2) return -1 ;				/* Y_SENTINEL untouched (pbwtCore.c:409) */
  for (int i = 0 ; i < nsteps ; ++i) pbwtCursorForwardsReadAD (u, k + i) ;
  memcpy (a_end, u->a, sizeof (int) * M) ; memcpy (d_end, u->d, sizeof (int) * (M + 1)) ;
  memcpy (y_end, u->y, M) ; *c_end = u->c ;
  pbwtCursorDestroy (u) ; pbwtDestroy (p) ;
  return 0 ;
}
