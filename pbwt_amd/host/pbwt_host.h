/* pbwt_host.h — host side (C) of the drop-in: the reference's file formats and command grammar for
 * the hot-path subset, with every whole-panel loop delegated to libpbwtgpu.so
 * through include/pbwt_amd.h.  Own code; mirrors the reference's behaviour (file:line cited at each
 * function) but shares no source with it.  No compute happens on the host: parsing, formatting and
 * the file formats only. */
#ifndef PBWT_HOST_H
#define PBWT_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "../../include/pbwt_amd.h"

typedef struct {
  int x ;			/* position (Site.x, pbwt.h:58) */
  char *var ;			/* variation text "REF\tALT" (variationDict entry), NULL if none */
} HostSite ;

typedef struct {		/* the fields of PBWT (pbwt.h:35-53) the hot path touches */
  int M, N ;
  char *chrom ;
  HostSite *sites ;		/* N entries or NULL */
  uint8_t *yz ; int64_t nz ;	/* packed columns (PBWT.yz) */
  int *aFstart, *aFend ;
  uint8_t *zz ; int64_t nzz ;	/* packed reverse PBWT (PBWT.zz) and its index arrays, or NULL */
  int *aRstart, *aRend ;
} Panel ;

extern FILE *logFile ;
extern int isCheck, isStats ;

void die (const char *format, ...) ;		/* utils.c:31 behaviour: message on stderr, exit(-1) */
Panel *panelCreate (int M, int N) ;		/* pbwtCreate, pbwtCore.c:41-49 */
void panelDestroy (Panel *p) ;
Panel *panelRead (FILE *fp) ;			/* pbwtRead, pbwtIO.c:172-217 */
void panelWrite (Panel *p, FILE *fp) ;		/* pbwtWrite, pbwtIO.c:33-57 */
void panelReadSites (Panel *p, FILE *fp) ;	/* pbwtReadSites, pbwtIO.c:232-276 */
void panelWriteSites (Panel *p, FILE *fp) ;	/* pbwtWriteSites, pbwtIO.c:59-77 */
Panel *panelReadAll (const char *root) ;	/* pbwtReadAll, pbwtIO.c:408-422 (.pbwt + .sites) */
void panelWriteAll (Panel *p, const char *root) ;	/* pbwtWriteAll, pbwtIO.c:134-144 */
extern int nCheckPoint ;			/* -checkpoint n: write check_A/check_B .pbwt + .sites every n sites while reading (pbwtIO.c:27,158-168) */
Panel *panelReadMacs (FILE *fp) ;		/* pbwtReadMacs, pbwtIO.c:426-492 */
void panelWriteHaplotypes (FILE *fp, Panel *p) ;	/* pbwtWriteHaplotypes, pbwtIO.c:839-857 */
void panelLongMatches (Panel *p, int L) ;	/* pbwtLongMatches, pbwtMatch.c:148-183 (L == 0: maximal) */
void panelMatchDynamic (Panel *p, FILE *fp) ;	/* matchSequencesDynamic, pbwtMatch.c:352-357 */
void panelSiteInfo (Panel *p, FILE *fp, int f1, int f2) ;	/* exportSiteInfo, pbwtMain.c:82-100 */
Panel *panelSubSampleInterval (Panel *p, int start, int Mnew) ;	/* pbwtSubSampleInterval, pbwtSample.c:95-108 */
Panel *panelSubSample (Panel *p, const int *select, int Mnew) ;	/* pbwtSubSample, pbwtSample.c:59-93 */
Panel *panelSubRange (Panel *p, int start, int end) ;		/* pbwtSubRange, pbwtCore.c:111-148 */
Panel *panelSelectSites (Panel *p, FILE *fp) ;			/* -selectSites: pbwtReadSitesFile + pbwtSelectSites, pbwtCore.c:623-682 */
Panel *panelRemoveSites (Panel *p, FILE *fp) ;			/* -removeSites: pbwtRemoveSites, pbwtCore.c:686-732 */
void panelBuildReverse (Panel *p) ;		/* pbwtBuildReverse, pbwtCore.c:151-191 */
void panelWriteReverse (Panel *p, FILE *fp) ;	/* pbwtWriteReverse, pbwtIO.c:121-132 */
void panelReadReverse (Panel *p, FILE *fp) ;	/* pbwtReadReverse, pbwtIO.c:392-404 */
void timeUpdate (FILE *f) ;			/* utils.c:173-198 */
#endif
