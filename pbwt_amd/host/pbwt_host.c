/* pbwt_host.c — see pbwt_host.h.  File formats + glue; all panel-wide computation is in libpbwtgpu. */
#include "pbwt_host.h"
#include <ctype.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

FILE *logFile ;
int isCheck = 0, isStats = 0 ;

void die (const char *format, ...)
{
  va_list args ;
  va_start (args, format) ;
  fprintf (stderr, "FATAL ERROR: ") ;
  vfprintf (stderr, format, args) ;
  fprintf (stderr, "\n") ;
  va_end (args) ;
  timeUpdate (stderr) ;
  exit (-1) ;
}

/* allocation accounting for the "Memory" field of the timing line (the reference counts myalloc'ed bytes, utils.c:46-73) */
static long totalAllocated = 0 ;

static void *xalloc (size_t n)
{ void *p = calloc (n ? n : 1, 1) ; if (!p) die ("out of memory allocating %zu bytes", n) ;
  totalAllocated += (long) n ; return p ;
}

static void *xrealloc (void *old, size_t oldBytes, size_t newBytes)
{ void *p = realloc (old, newBytes ? newBytes : 1) ; if (!p) die ("out of memory growing a buffer to %zu bytes", newBytes) ;
  totalAllocated += (long) newBytes - (long) oldBytes ; return p ;
}

/* the timing line after every command, in the reference's format (utils.c:173-198):
   user\t<s.us>\tsystem\t<s.us>\tmax_RSS\t<growth since the last line>\tMemory\t<bytes allocated> */
void timeUpdate (FILE *f)
{
  static int started = 0 ;
  static struct rusage last ;
  struct rusage now ;
  getrusage (RUSAGE_SELF, &now) ;
  if (started)
    { const struct timeval *t1[2] = { &now.ru_utime, &now.ru_stime }, *t0[2] = { &last.ru_utime, &last.ru_stime } ;
      const char *label[2] = { "user", "\tsystem" } ;
      for (int i = 0 ; i < 2 ; ++i)
	{ long us = (long) (t1[i]->tv_sec - t0[i]->tv_sec) * 1000000L + (t1[i]->tv_usec - t0[i]->tv_usec) ;
	  fprintf (f, "%s\t%d.%06d", label[i], (int) (us / 1000000L), (int) (us % 1000000L)) ;
	}
      fprintf (f, "\tmax_RSS\t%ld\tMemory\t%li\n", now.ru_maxrss - last.ru_maxrss, totalAllocated) ;
    }
  started = 1 ;
  last = now ;
}

/* one engine per panel width, created on first use; failure is fatal (no CPU fallback) */
static pbwtamd_engine *engineFor (int M)
{
  static pbwtamd_engine *e = 0 ; static int Mcur = 0 ;
  if (e && Mcur != M) { pbwtamd_engine_destroy (e) ; e = 0 ; }
  if (!e && pbwtamd_engine_create (&e, 0, M, 0, 0)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  Mcur = M ;
  return e ;
}

Panel *panelCreate (int M, int N)
{
  Panel *p = xalloc (sizeof (Panel)) ;
  if (M <= 0 || N < 0) die ("pbwtCreate called with bad M %d or N %d", M, N) ;
  p->M = M ; p->N = N ;
  p->aFstart = xalloc (sizeof (int) * M) ;
  for (int i = 0 ; i < M ; ++i) p->aFstart[i] = i ;
  return p ;
}

void panelDestroy (Panel *p)
{
  if (!p) return ;
  if (p->sites) { for (int i = 0 ; i < p->N ; ++i) free (p->sites[i].var) ; free (p->sites) ; }
  free (p->chrom) ; free (p->yz) ; free (p->aFstart) ; free (p->aFend) ; free (p->zz) ; free (p->aRstart) ; free (p->aRend) ; free (p) ;
}

/* .pbwt: "PBW3", int M, int N, aFstart[M], aFend[M], long nz, 4 pad bytes, yz[nz] (pbwtIO.c:33-57).
 * Reader also takes PBW2 (int nz, no pad), PBWT (no index arrays) and GBWT (pbwtIO.c:182-208). */
void panelWrite (Panel *p, FILE *fp)
{
  if (!p || !p->yz) die ("pbwtWrite called without a valid pbwt") ;
  if (!p->aFstart || !p->aFend) die ("pbwtWrite called without start and end indexes") ;
  long n = (long) p->nz ;
  if (fwrite ("PBW3", 1, 4, fp) != 4 || fwrite (&p->M, sizeof (int), 1, fp) != 1 || fwrite (&p->N, sizeof (int), 1, fp) != 1
      || fwrite (p->aFstart, sizeof (int), p->M, fp) != (size_t) p->M || fwrite (p->aFend, sizeof (int), p->M, fp) != (size_t) p->M
      || fwrite (&n, sizeof (long), 1, fp) != 1 || fwrite ("    ", 1, 4, fp) != 4
      || fwrite (p->yz, 1, (size_t) n, fp) != (size_t) n)
    die ("error writing PBWT in pbwtWrite") ;
  fprintf (logFile, "written %ld chars pbwt: M, N are %d, %d\n", n, p->M, p->N) ;
}

/* the four on-disk variants differ in two things only: whether the index arrays are stored and how wide the byte count is */
static const struct { const char *tag ; int hasIndex, wideCount ; } pbwtKinds[] =
  { { "PBW3", 1, 1 }, { "PBW2", 1, 0 }, { "PBWT", 0, 0 }, { "GBWT", 0, 0 } } ;

static void readOrDie (void *dst, size_t size, size_t n, FILE *fp, const char *what)
{ if (fread (dst, size, n, fp) != n) die ("error reading %s in pbwtRead", what) ; }

Panel *panelRead (FILE *fp)
{
  struct { char tag[4] ; int32_t M, N ; } head ;		/* 12 bytes, no padding */
  if (fread (head.tag, 1, 4, fp) != 4) die ("failed to read 4 char tag - is file readable?") ;
  int kind = -1 ;
  for (int i = 0 ; i < 4 ; ++i) if (!memcmp (head.tag, pbwtKinds[i].tag, 4)) kind = i ;
  if (kind < 0) die ("failed to recognise file type %.4s in pbwtRead - was it written by pbwt?", head.tag) ;
  readOrDie (&head.M, sizeof (int32_t), 1, fp, "m") ;
  readOrDie (&head.N, sizeof (int32_t), 1, fp, "n") ;
  Panel *p = panelCreate (head.M, head.N) ;
  if (pbwtKinds[kind].hasIndex)
    { p->aFend = xalloc (sizeof (int) * (size_t) p->M) ;
      readOrDie (p->aFstart, sizeof (int), (size_t) p->M, fp, "aFstart") ;
      readOrDie (p->aFend, sizeof (int), (size_t) p->M, fp, "aFend") ;
    }
  if (pbwtKinds[kind].wideCount)
    { struct { int64_t n ; char pad[4] ; } cnt ;
      if (fread (&cnt.n, 8, 1, fp) != 1 || fread (cnt.pad, 1, 4, fp) != 4) die ("error reading pbwt file") ;
      p->nz = cnt.n ;
    }
  else
    { int32_t n32 ; if (fread (&n32, 4, 1, fp) != 1) die ("error reading pbwt file") ; p->nz = n32 ; }
  if (p->nz < 0) die ("error reading pbwt file") ;
  p->yz = xalloc ((size_t) p->nz) ;
  if (fread (p->yz, 1, (size_t) p->nz, fp) != (size_t) p->nz) die ("error reading data in pbwt file") ;
  fprintf (logFile, "read pbwt %.4s file with %ld bytes: M, N are %d, %d\n", head.tag, (long) p->nz, p->M, p->N) ;
  return p ;
}

/* .sites: "chrom\tpos\tvariation" per site (pbwtIO.c:59-77); glibc prints a NULL variation as
 * "(null)", which is what the reference emits for MaCS panels (no variation dictionary entry) */
void panelWriteSites (Panel *p, FILE *fp)
{
  if (!p || !p->sites) die ("pbwtWriteSites called without sites") ;
  for (int i = 0 ; i < p->N ; ++i)
    fprintf (fp, "%s\t%d\t%s\n", p->chrom ? p->chrom : ".", p->sites[i].x, p->sites[i].var ? p->sites[i].var : "(null)") ;
  if (ferror (fp)) die ("error writing sites file") ;
  fprintf (logFile, "written %d sites from %d to %d\n", p->N, p->sites[0].x, p->sites[p->N-1].x) ;
}

void panelReadSites (Panel *p, FILE *fp)
{
  if (!p) die ("pbwtReadSites called without a valid pbwt") ;
  size_t cap = 4096, n = 0, len = 0 ;
  HostSite *sites = xalloc (cap * sizeof (HostSite)) ;
  char *line = 0 ;
  ssize_t got ;
  int lineNo = 1 ;
  while ((got = getline (&line, &len, fp)) > 0)
    { char *s = line ;
      while (got > 0 && (s[got-1] == '\n' || s[got-1] == '\r')) s[--got] = 0 ;
      if (!got) continue ;
      char *tab = strchr (s, '\t') ; if (!tab) tab = strchr (s, ' ') ;
      if (!tab) die ("bad position line %d in sites file", lineNo) ;
      *tab = 0 ;
      if (strcmp (s, "."))		/* readMatchChrom (pbwtIO.c:219-230): match if set, else adopt */
	{ if (p->chrom && strcmp (p->chrom, s)) die ("failed to match chromosome in sites file: line %d", lineNo) ;
	  if (!p->chrom) p->chrom = strdup (s) ;
	}
      char *q = tab + 1 ;
      if (!isdigit ((unsigned char) *q)) die ("bad position line %d in sites file", lineNo) ;
      if (n == cap) { sites = xrealloc (sites, cap * sizeof (HostSite), 2 * cap * sizeof (HostSite)) ; cap *= 2 ; }
      sites[n].x = 0 ; while (isdigit ((unsigned char) *q)) sites[n].x = sites[n].x * 10 + (*q++ - '0') ;
      sites[n].var = 0 ;
      if (*q) { while (*q && isspace ((unsigned char) *q)) ++q ; if (*q) sites[n].var = strdup (q) ; }
      ++n ; ++lineNo ;
    }
  free (line) ;
  if (ferror (fp)) die ("error reading sites file") ;
  /* the reference's reader runs its chromosome match once more at end of file (pbwtIO.c:240-241 with fgetword returning ""
     there), which leaves a panel whose sites all say "." with chrom = "" rather than unset: -writeSites then prints an
     empty first column.  Kept, so that files round-trip byte for byte. */
  if (!p->chrom) p->chrom = strdup ("") ;
  fprintf (logFile, "read %zu sites on chromosome %s from file\n", n, p->chrom) ;
  if ((int) n != p->N) die ("sites file contains %zu sites not %d as in pbwt", n, p->N) ;
  p->sites = sites ;
}

static FILE *fopenTag (const char *root, const char *tag, const char *mode)
{ char *name = xalloc (strlen (root) + strlen (tag) + 2) ;
  sprintf (name, "%s.%s", root, tag) ;
  FILE *f = fopen (name, mode) ; free (name) ; return f ;
}

void panelWriteAll (Panel *p, const char *root)
{
  FILE *fp ;
  if (!(fp = fopenTag (root, "pbwt", "w"))) die ("failed to open root.%s", "pbwt") ;
  panelWrite (p, fp) ; fclose (fp) ;
  if (p->sites) { if (!(fp = fopenTag (root, "sites", "w"))) die ("failed to open root.%s", "sites") ; panelWriteSites (p, fp) ; fclose (fp) ; }
  if (p->zz) { if (!(fp = fopenTag (root, "reverse", "w"))) die ("failed to open root.%s", "reverse") ; panelWriteReverse (p, fp) ; fclose (fp) ; }	/* pbwtIO.c:143 */
}

Panel *panelReadAll (const char *root)
{
  FILE *fp ; Panel *p ;
  if ((fp = fopenTag (root, "pbwt", "r"))) { p = panelRead (fp) ; fclose (fp) ; }
  else die ("failed to open %s.pbwt", root) ;
  if ((fp = fopenTag (root, "sites", "r"))) { panelReadSites (p, fp) ; fclose (fp) ; }
  if ((fp = fopenTag (root, "reverse", "r"))) { panelReadReverse (p, fp) ; fclose (fp) ; }	/* pbwtIO.c:419 */
  return p ;
}

/* MaCS text (pbwtIO.c:426-458): "COMMAND: <cmd> M L ...", "SEED: ...", then "SITE: n pos time <M chars>".
 * The parsed alleles go straight into bit columns; the per-site gather/pack3/partition loop of
 * pbwtIO.c:477-483 runs on the device (pbwtamd_build). */
static char *word (FILE *fp, char *buf, int cap)
{ int c, n = 0 ;
  while ((c = getc (fp)) != EOF && isgraph (c)) if (n < cap - 1) buf[n++] = (char) c ;
  while (c != EOF && c != '\n' && !isgraph (c)) c = getc (fp) ;
  if (c != EOF) ungetc (c, fp) ;
  buf[n] = 0 ; return buf ;
}

int nCheckPoint = 0 ;

/* extend the panel by the parsed columns [*built, n): the device build continues from the cursor the previous chunk
 * ended with (aFend), the new pack3 bytes are appended (runs never span columns) */
static void buildMore (Panel *p, pbwtamd_engine *e, const uint32_t *cols, int wpc, int n, int *built)
{
  if (!p->aFend) p->aFend = xalloc (sizeof (int) * p->M) ;
  if (n > *built || !*built)
    { uint8_t *yz = 0 ; int64_t nz = 0 ;
      int *start = *built ? xalloc (sizeof (int) * p->M) : p->aFstart ;
      if (*built) memcpy (start, p->aFend, sizeof (int) * p->M) ;
      if (pbwtamd_build (e, cols + (size_t) *built * wpc, wpc, n - *built, 0, start, &yz, &nz, p->aFend, 0)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
      if (*built) free (start) ;
      if (!p->yz) { p->yz = yz ; p->nz = nz ; }
      else
	{ p->yz = xrealloc (p->yz, (size_t) p->nz, (size_t) (p->nz + nz) + 1) ;
	  memcpy (p->yz + p->nz, yz, (size_t) nz) ; p->nz += nz ; pbwtamd_free (yz) ;
	}
    }
  p->N = n ; *built = n ;
}

Panel *panelReadMacs (FILE *fp)
{
  char w[256] ;
  int c ;
  if (strcmp (word (fp, w, 256), "COMMAND:")) die ("MaCS COMMAND line not found") ;
  word (fp, w, 256) ;
  int M = atoi (word (fp, w, 256)) ; if (!M) die ("failed to get M") ;
  double L = atof (word (fp, w, 256)) ; if (!L) die ("failed to get L") ;
  while ((c = getc (fp)) != '\n' && c != EOF) ;
  if (strcmp (word (fp, w, 256), "SEED:")) die ("SEED line not found") ;
  while ((c = getc (fp)) != '\n' && c != EOF) ;

  Panel *p = panelCreate (M, 0) ;
  pbwtamd_engine *e = engineFor (M) ;
  const int wpc = pbwtamd_engine_wpc (e) ;
  size_t cap = 1024, n = 0 ;
  int built = 0 ;
  uint32_t *cols = xalloc (cap * wpc * sizeof (uint32_t)) ;
  p->sites = xalloc (cap * sizeof (HostSite)) ;
  while (!feof (fp) && !strcmp (word (fp, w, 256), "SITE:"))
    { if (n == cap)
	{ cols = xrealloc (cols, cap * wpc * sizeof (uint32_t), 2 * cap * wpc * sizeof (uint32_t)) ;
	  p->sites = xrealloc (p->sites, cap * sizeof (HostSite), 2 * cap * sizeof (HostSite)) ;
	  cap *= 2 ;
	}
      int number = atoi (word (fp, w, 256)) ;
      p->sites[n].x = (int) (L * atof (word (fp, w, 256))) ; p->sites[n].var = 0 ;
      word (fp, w, 256) ;				/* the time, ignored */
      uint32_t *col = cols + n * wpc ;
      memset (col, 0, wpc * sizeof (uint32_t)) ;
      for (int h = 0 ; h < M ; ++h) if (getc (fp) == '1') col[h >> 5] |= 1u << (h & 31) ;
      if (feof (fp)) break ;
      if (getc (fp) != '\n') die ("end of line error for MaCS SITE %d", number) ;
      ++n ;
      if (nCheckPoint && !(n % nCheckPoint))		/* pbwtIO.c:481 -> pbwtCheckPoint (pbwtIO.c:158-168) */
	{ static int isA = 1 ;
	  buildMore (p, e, cols, wpc, (int) n, &built) ;
	  panelWriteAll (p, isA ? "check_A" : "check_B") ;
	  isA = !isA ;
	}
    }
  buildMore (p, e, cols, wpc, (int) n, &built) ;
  free (cols) ;
  fprintf (logFile, "read MaCS file: M, N are\t%d\t%d\n", M, p->N) ;
  return p ;
}

/* -haps (pbwtWriteHaplotypes, pbwtIO.c:839-857): the device hands over the alleles a batch of sites at a time */
typedef struct { FILE *fp ; int M ; char *line ; } HapSink ;
static void writeHapRows (int k0, int nsites, const uint8_t *rows, void *ctx)
{ HapSink *h = (HapSink*) ctx ; (void) k0 ;
  for (int i = 0 ; i < nsites ; ++i)
    { for (int j = 0 ; j < h->M ; ++j) h->line[j] = rows[(size_t) i * h->M + j] ? '1' : '0' ;
      h->line[h->M] = '\n' ; fwrite (h->line, 1, (size_t) h->M + 1, h->fp) ;
    }
}

void panelWriteHaplotypes (FILE *fp, Panel *p)
{
  HapSink h = { fp, p->M, xalloc ((size_t) p->M + 2) } ;
  if (pbwtamd_haplotypes_stream (engineFor (p->M), p->yz, p->nz, p->N, p->aFstart, writeHapRows, &h)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  free (h.line) ;
  fprintf (logFile, "written haplotype file: %d rows of %d\n", p->N, p->M) ;
}

/* reportMatch (pbwtMatch.c:46-58); -check (checkMatchMaximal, :33-44) against the decoded panel */
static uint8_t *checkA, *checkB ; static int checkMA, checkMB, checkN ;
static void reportMatch (int ai, int bi, int start, int end)
{
  if (start == end) return ;
  /* the text of printf ("MATCH\t%d\t%d\t%d\t%d\t%d\n", ...) (pbwtMatch.c:48), digits by hand: configs[1] prints 10^8 of these lines, and the format parser was
     most of what `pbwt -read f -maxWithin` spent on the host */
  { char buf[72] ; char *q = buf + sizeof buf ; int v[5] = { ai, bi, start, end, end - start } ;
    *--q = '\n' ;
    for (int f = 4 ; f >= 0 ; --f)
      { unsigned u = v[f] < 0 ? 0u - (unsigned) v[f] : (unsigned) v[f] ;
	do { *--q = (char) ('0' + u % 10) ; u /= 10 ; } while (u) ;
	if (v[f] < 0) *--q = '-' ;
	*--q = '\t' ;
      }
    q -= 5 ; memcpy (q, "MATCH", 5) ;
    fwrite (q, 1, (size_t) (buf + sizeof buf - q), stdout) ;
  }
  if (isCheck)
    { (void) checkMA ; (void) checkMB ;
#define HA(k) checkA[(size_t)(k) * checkMA + ai]
#define HB(k) checkB[(size_t)(k) * checkMB + bi]
      if (start && HA(start-1) == HB(start-1)) die ("match not maximal - can extend backwards\n") ;
      if (end < checkN && HA(end) == HB(end)) die ("match not maximal - can extend forwards\n") ;
      for (int i = start ; i < end ; ++i) if (HA(i) != HB(i)) die ("match not a match at %d\n", i) ;
    }
}

static uint8_t *decodeHaps (Panel *p)
{ uint8_t *hap = xalloc ((size_t) p->N * p->M) ;
  if (pbwtamd_haplotypes (engineFor (p->M), p->yz, p->nz, p->N, p->aFstart, hap)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  return hap ;
}

void panelLongMatches (Panel *p, int L)
{
  if (!p || !p->yz) die ("option -longWithin called without a PBWT") ;
  if (L < 0) die ("L %d for longWithin must be >= 0", L) ;
  pbwtamd_engine *e = engineFor (p->M) ;
  if (isCheck) { checkA = checkB = decodeHaps (p) ; checkMA = checkMB = p->M ; checkN = p->N ; }
  int64_t *h = isStats ? xalloc (sizeof (int64_t) * ((size_t) p->N + 1)) : 0 ;	/* matchLengthHist (pbwtMatch.c:158-159) */
  if (L)				/* matchLongWithin2 (pbwtMatch.c:85-113): reports even under -stats, fills no histogram */
    { if (pbwtamd_long_within (e, p->yz, p->nz, p->N, p->aFstart, L, reportMatch, 0, 0)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ; }
  else if (isStats)			/* histogram instead of reports (pbwtMatch.c:130-131) */
    { if (pbwtamd_max_within (e, p->yz, p->nz, p->N, p->aFstart, 0, 0, 0, h, p->N + 1)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ; }
  else if (pbwtamd_max_within (e, p->yz, p->nz, p->N, p->aFstart, reportMatch, 0, 0, 0, 0)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  if (isStats)				/* pbwtMatch.c:166-178, also after -longWithin (all-zero histogram, 0/0 average) */
    { long nTot = 0, hTot = 0 ;
      for (int i = 0 ; i <= p->N ; ++i)
	if (h[i]) { nTot += h[i] ; hTot += h[i] * i ; printf ("%d\t%ld\n", i, (long) h[i]) ; }
      fprintf (logFile, "Average %.1f matches per sample\n", nTot / (double) p->M) ;
      fprintf (logFile, "Average length %.1f\n", hTot / (double) nTot) ;
      free (h) ;
    }
  if (isCheck) { free (checkA) ; checkA = checkB = 0 ; }
}

void panelMatchDynamic (Panel *p, FILE *fp)
{
  Panel *q = panelRead (fp) ;
  if (q->N != p->N) die ("query length in matchSequences %d != PBWT length %d", q->N, p->N) ;
  if (isCheck) { checkA = decodeHaps (q) ; checkB = decodeHaps (p) ; checkMA = q->M ; checkMB = p->M ; checkN = p->N ; }
  int64_t nomatch = 0, tot[2] = {0, 0} ;
  if (pbwtamd_match_sweep (engineFor (p->M), p->yz, p->nz, p->N, p->aFstart, q->M, q->yz, q->nz, q->aFstart,
			   reportMatch, 0, 0, &nomatch, tot))
    die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  if (nomatch)				/* pbwtMatch.c:405-410, one line per event in the reference's order */
    { int32_t *ev = 0 ; int64_t nev = 0 ;
      if (pbwtamd_get_nomatch_events (engineFor (p->M), &ev, &nev)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
      for (int64_t i = 0 ; i < nev ; ++i)
	fprintf (logFile, "no match to query %d value %d at site %d\n", ev[4*i], ev[4*i+1], ev[4*i+2]) ;
      if (nev < nomatch)		/* the library keeps a bounded number of events (the count is exact) */
	fprintf (logFile, "... %lld further no-match events not listed\n", (long long) (nomatch - nev)) ;
      pbwtamd_free (ev) ;
    }
  fprintf (logFile, "Average number of best matches including alternates %.1f, Average length %.1f, Av number per position %.1f\n",
	   tot[0] / (double) q->M, tot[1] / (double) tot[0], tot[1] / (double) ((long) q->M * q->N)) ;
  if (isCheck) { free (checkA) ; free (checkB) ; checkA = checkB = 0 ; }
  panelDestroy (q) ;
}

/* reverse PBWT (pbwtCore.c:151-191): the same build loop over the sites in reverse order, started from the forward
 * pass's final order aFend — decode, regather and rebuild all on the device (pbwtamd_regather) */
void panelBuildReverse (Panel *p)
{
  pbwtamd_engine *e = engineFor (p->M) ;
  int *order = xalloc (sizeof (int) * ((size_t) p->N + 1)) ;
  for (int k = 0 ; k < p->N ; ++k) order[k] = p->N - 1 - k ;	/* column k of the reverse panel = site N-1-k */
  if (!p->aFend)			/* run forwards to the end first (pbwtCore.c:160-165) */
    { p->aFend = xalloc (sizeof (int) * p->M) ;
      int last = p->N ;
      if (pbwtamd_sweep_AD (e, p->yz, p->nz, p->N, p->aFstart, 0, 0, 0, &last, 1, p->aFend, 0, 0)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
    }
  free (p->zz) ; free (p->aRstart) ; free (p->aRend) ;
  p->aRstart = xalloc (sizeof (int) * p->M) ; memcpy (p->aRstart, p->aFend, sizeof (int) * p->M) ;
  p->aRend = xalloc (sizeof (int) * p->M) ;
  if (pbwtamd_regather (e, p->yz, p->nz, p->N, p->aFstart, order, p->N, 0, p->M, p->aRstart, &p->zz, &p->nzz, p->aRend, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  free (order) ;
  fprintf (logFile, "built reverse PBWT - size %ld\n", (long) p->nzz) ;
}

void panelWriteReverse (Panel *p, FILE *fp)
{
  if (!p || !p->zz) die ("pbwtWriteReverse called without reverse pbwt") ;
  Panel q = *p ;
  q.yz = p->zz ; q.nz = p->nzz ; q.aFstart = p->aRstart ; q.aFend = p->aRend ;
  fprintf (logFile, "reverse: ") ; panelWrite (&q, fp) ;
}

void panelReadReverse (Panel *p, FILE *fp)
{
  if (!p) die ("pbwtReadReverse called without a valid pbwt") ;
  Panel *q = panelRead (fp) ;
  if (q->M != p->M || q->N != p->N) die ("M %d or N %d in reverse don't match %d, %d in forward", q->M, q->N, p->M, p->N) ;
  free (p->zz) ; free (p->aRstart) ; free (p->aRend) ;
  p->zz = q->yz ; p->nzz = q->nz ; q->yz = 0 ;
  p->aRstart = q->aFstart ; q->aFstart = 0 ;
  p->aRend = q->aFend ; q->aFend = 0 ;
  panelDestroy (q) ;
}

/* number of ones in each packed column: a format-level scan of the run bytes (pbwtCore.c:216-225) */
static int *onesPerColumn (Panel *p)
{
  int *ones = xalloc (sizeof (int) * ((size_t) p->N + 1)) ;
  int64_t b = 0 ;
  for (int k = 0 ; k < p->N ; ++k)
    { int m = 0 ;
      while (m < p->M && b < p->nz)
	{ uint8_t z = p->yz[b++] ; int v = z & 0x7f ;
	  int n = v < 64 ? v : (v < 96 ? (v - 64) << 6 : (v - 96) << 11) ;
	  if (z & 0x80) ones[k] += n ;
	  m += n ;
	}
    }
  return ones ;
}

void panelSiteInfo (Panel *p, FILE *fp, int f1, int f2)
{
  int *ones = onesPerColumn (p), n = 0 ;
  int *sel = xalloc (sizeof (int) * ((size_t) p->N + 1)) ;
  for (int i = 0 ; i < p->N ; ++i) if (f1 <= ones[i] && ones[i] < f2) sel[n++] = i ;
  if (n)
    { int *a = xalloc (sizeof (int) * (size_t) n * p->M), *d = xalloc (sizeof (int) * (size_t) n * (p->M + 1)) ;
      uint8_t *y = xalloc ((size_t) n * p->M) ;
      if (pbwtamd_sweep_AD (engineFor (p->M), p->yz, p->nz, p->N, p->aFstart, 0, 0, 0, sel, n, a, d, y)) die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
      for (int r = 0 ; r < n ; ++r)
	{ for (int j = 0 ; j < p->M ; ++j) fprintf (fp, "%d %d ", y[(size_t) r * p->M + j], sel[r] - d[(size_t) r * (p->M + 1) + j]) ;
	  fprintf (fp, "\n") ;
	}
      free (a) ; free (d) ; free (y) ;
    }
  free (ones) ; free (sel) ;
  fprintf (logFile, "%d rows exported with allele count f, %d <= f < %d\n", n, f1, f2) ;
}

/* site i of p -> a fresh copy for another panel */
static HostSite copySite (const HostSite *s)
{ HostSite c ; c.x = s->x ; c.var = s->var ? strdup (s->var) : 0 ; return c ; }

/* the shared tail of the panel transforms: a new panel of Mnew haplotypes (select[h] of the old ones, or all) over the
 * sites order[0..nOut) of the old one, rebuilt on the device; consumes p like the reference's transforms do */
static Panel *regathered (Panel *p, const int *order, int nOut, const int *select, int Mnew)
{
  Panel *q = panelCreate (Mnew, nOut) ;
  q->aFend = xalloc (sizeof (int) * Mnew) ;
  if (pbwtamd_regather (engineFor (p->M), p->yz, p->nz, p->N, p->aFstart, order, nOut, select, Mnew, q->aFstart, &q->yz, &q->nz, q->aFend, 0))
    die ("pbwt_amd: %s", pbwtamd_last_error ()) ;
  if (p->chrom) q->chrom = strdup (p->chrom) ;
  if (p->sites)
    { q->sites = xalloc (sizeof (HostSite) * ((size_t) nOut + 1)) ;
      for (int i = 0 ; i < nOut ; ++i) q->sites[i] = copySite (&p->sites[order ? order[i] : i]) ;
    }
  panelDestroy (p) ;
  return q ;
}

/* pbwtSubSample (pbwtSample.c:59-93): select[i] is the position in old of the i'th haplotype in new */
Panel *panelSubSample (Panel *p, const int *select, int Mnew)
{
  if (!p || !p->yz) die ("subSample called without valid pbwt") ;
  return regathered (p, 0, p->N, select, Mnew) ;
}

Panel *panelSubSampleInterval (Panel *p, int start, int Mnew)	/* pbwtSample.c:95-108 */
{
  if (start < 0 || Mnew <= 0 || start + Mnew > p->M) die ("bad start %d, Mnew %d in subsample", start, Mnew) ;
  int *select = xalloc (sizeof (int) * (size_t) Mnew) ;
  for (int i = 0 ; i < Mnew ; ++i) select[i] = start + i ;
  Panel *q = panelSubSample (p, select, Mnew) ;
  free (select) ;
  return q ;
}

Panel *panelSubRange (Panel *p, int start, int end)		/* pbwtSubRange, pbwtCore.c:111-148 */
{
  if (!p || !p->yz) die ("subrange without an existing pbwt") ;
  if (start < 0 || end > p->N || end <= start) die ("subrange invalid start %d, end %d", start, end) ;
  int *order = xalloc (sizeof (int) * (size_t) (end - start)) ;
  for (int i = start ; i < end ; ++i) order[i - start] = i ;
  Panel *q = regathered (p, order, end - start, 0, p->M) ;
  free (order) ;
  return q ;
}

/* a sites file as a list (pbwtReadSitesFile, pbwtIO.c:232-267) */
static HostSite *readSitesList (FILE *fp, char **chrom, int *n)
{
  Panel tmp ; memset (&tmp, 0, sizeof tmp) ;
  size_t cap = 1024, cnt = 0, len = 0 ; char *line = 0 ; ssize_t got ;
  HostSite *v = xalloc (cap * sizeof (HostSite)) ;
  while ((got = getline (&line, &len, fp)) > 0)
    { while (got > 0 && (line[got-1] == '\n' || line[got-1] == '\r')) line[--got] = 0 ;
      if (!got) continue ;
      char *tab = strchr (line, '\t') ; if (!tab) tab = strchr (line, ' ') ;
      if (!tab) die ("bad position line %zu in sites file", cnt + 1) ;
      *tab = 0 ;
      if (strcmp (line, "."))				/* readMatchChrom, pbwtIO.c:219-230 */
	{ if (!*chrom) *chrom = strdup (line) ;
	  else if (strcmp (*chrom, line)) die ("failed to match chromosome in sites file: line %zu", cnt + 1) ;
	}
      char *q = tab + 1 ;
      if (!isdigit ((unsigned char) *q)) die ("bad position line %zu in sites file", cnt + 1) ;
      if (cnt == cap) { v = xrealloc (v, cap * sizeof (HostSite), 2 * cap * sizeof (HostSite)) ; cap *= 2 ; }
      v[cnt].x = 0 ; while (isdigit ((unsigned char) *q)) v[cnt].x = v[cnt].x * 10 + (*q++ - '0') ;
      while (*q && isspace ((unsigned char) *q)) ++q ;
      v[cnt].var = *q ? strdup (q) : 0 ;
      ++cnt ;
    }
  free (line) ;
  if (!*chrom) *chrom = strdup ("") ;			/* as panelReadSites: the reference's end-of-file chromosome match */
  *n = (int) cnt ;
  fprintf (logFile, "read %ld sites on chromosome %s from file\n", (long) cnt, *chrom) ;	/* pbwtIO.c:263 */
  return v ;
}

/* the reference orders variations by their index in the global variation dictionary (first-come order of the strings it has
 * seen: the panel's sites first, then the list's): the same numbering here */
typedef struct { char **name ; int n, cap ; } VarDict ;
static int varIndex (VarDict *d, const char *s)
{ if (!s) s = "" ;
  for (int i = 0 ; i < d->n ; ++i) if (!strcmp (d->name[i], s)) return i ;
  if (d->n == d->cap) { d->name = xrealloc (d->name, sizeof (char*) * (size_t) d->cap, sizeof (char*) * (size_t) (2 * d->cap + 16)) ; d->cap = 2 * d->cap + 16 ; }
  d->name[d->n] = strdup (s) ; return d->n++ ;
}
static int noAlt (const char *s) { size_t n = s ? strlen (s) : 0 ; return n && s[n-1] == '.' ; }

/* pbwtSelectSites / pbwtRemoveSites (pbwtCore.c:623-732): the merge of the panel's sites with the list by position, then
 * by variation unless one of the two has no ALT; keep (or drop) the matches */
static Panel *selectOrRemove (Panel *p, FILE *fp, int keepMatches)
{
  if (!p || !p->sites) die ("%s called without sites", keepMatches ? "selectSites" : "removeSites") ;
  char *chr = 0 ; int nList = 0 ;
  HostSite *list = readSitesList (fp, &chr, &nList) ;
  if (p->chrom && chr && strcmp (chr, p->chrom)) die ("chromosome mismatch in %s", keepMatches ? "selectSites" : "removeSites") ;
  VarDict dict = { 0, 0, 0 } ;
  int *vp = xalloc (sizeof (int) * ((size_t) p->N + 1)), *vl = xalloc (sizeof (int) * ((size_t) nList + 1)) ;
  for (int i = 0 ; i < p->N ; ++i) vp[i] = varIndex (&dict, p->sites[i].var) ;
  for (int i = 0 ; i < nList ; ++i) vl[i] = varIndex (&dict, list[i].var) ;
  int *order = xalloc (sizeof (int) * ((size_t) p->N + 1)), nOut = 0, ip = 0, ia = 0 ;
  while (ip < p->N && ia < nList)
    { if (p->sites[ip].x < list[ia].x) { if (!keepMatches) order[nOut++] = ip ; ++ip ; }
      else if (p->sites[ip].x > list[ia].x) ++ia ;
      else
	{ int na = keepMatches && (noAlt (list[ia].var) || noAlt (p->sites[ip].var)) ;	/* pbwtRemoveSites compares the variations always */
	  if (!na && vp[ip] < vl[ia]) { if (!keepMatches) order[nOut++] = ip ; ++ip ; }
	  else if (!na && vp[ip] > vl[ia]) ++ia ;
	  else { if (keepMatches) order[nOut++] = ip ; ++ip ; ++ia ; }
	}
    }
  /* (pbwtRemoveSites stops with the list, pbwtCore.c:700: panel sites beyond the list's last position are not carried over) */
  const int Nold = p->N, M = p->M ;
  Panel *q = regathered (p, order, nOut, 0, M) ;
  fprintf (logFile, "%d sites selected from %d, pbwt size for %d haplotypes is %ld\n", nOut, Nold, M, (long) q->nz) ;
  for (int i = 0 ; i < nList ; ++i) free (list[i].var) ;
  for (int i = 0 ; i < dict.n ; ++i) free (dict.name[i]) ;
  free (dict.name) ; free (list) ; free (chr) ; free (vp) ; free (vl) ; free (order) ;
  return q ;
}

Panel *panelSelectSites (Panel *p, FILE *fp) { return selectOrRemove (p, fp, 1) ; }
Panel *panelRemoveSites (Panel *p, FILE *fp) { return selectOrRemove (p, fp, 0) ; }
