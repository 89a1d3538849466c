/* pbwtMatchGpu.c — unity build of the reference's pbwtMatch.c with the whole-panel loops of the hot path replaced by
 * the MI355X engine, WITHOUT editing the reference.  Compile INSTEAD of pbwtMatch.c:
 *     gcc -c -I<reference> -I<repo>/include integration/pbwtMatchGpu.c
 * pbwtLongMatches, reportMatch, checkMatchMaximal (-check), matchSequencesDynamic and the indexed matchers are the
 * reference's own code, compiled here from where it lies; their calls now reach the device.
 *
 * How: each replaced function becomes a function-like macro while pbwtMatch.c is included.  Its first argument tells
 * a DEFINITION (`PBWT *p, ...`: first token PBWT) from a CALL (`p, ...`, the spelling of every call site in
 * pbwtMatch.c:162,164,355); token pasting picks `name_cpu (PBWT` for the former — the CPU body stays in the object
 * under the _cpu name, unreferenced — and `name (p` for the latter (not re-expanded: a macro is never expanded inside
 * its own expansion), i.e. a call of the real name, which pbwtGpu.c defines below.
 * A maintainer who can edit pbwtMatch.c wraps the four bodies in `#ifndef PBWT_GPU` and adds pbwtGpu.o to the link
 * instead; this file exists so that the binding is compiled and tested against an untouched tree. */
/* pbwt.h has no include guard, so it cannot come first: declare the real names on an incomplete PBWT (C11 lets
   pbwt.h:35-53 repeat the typedef) before the macros below turn pbwt.h's own prototypes into _cpu ones */
struct PBWTstruct ; typedef struct PBWTstruct PBWT ;
void matchMaximalWithin (PBWT *p, void (*report)(int ai, int bi, int start, int end)) ;
void matchSequencesSweep (PBWT *p, PBWT *q, void (*report)(int ai, int bi, int start, int end)) ;
static void matchLongWithin2 (PBWT *p, int T, void (*report)(int ai, int bi, int start, int end)) ;

#define matchMaximalWithin(a, ...)        PBWTGPU_MMW_##a, __VA_ARGS__)
#define PBWTGPU_MMW_p                     matchMaximalWithin (p
#define PBWTGPU_MMW_PBWT                  matchMaximalWithin_cpu (PBWT
#define matchLongWithin2(a, ...)          PBWTGPU_MLW_##a, __VA_ARGS__)
#define PBWTGPU_MLW_p                     matchLongWithin2 (p
#define PBWTGPU_MLW_PBWT                  matchLongWithin2_cpu (PBWT
#define matchSequencesSweep(a, ...)       PBWTGPU_MSS_##a, __VA_ARGS__)
#define PBWTGPU_MSS_p                     matchSequencesSweep (p
#define PBWTGPU_MSS_PBWT                  matchSequencesSweep_cpu (PBWT
#define matchSequencesSweepSparse(a, ...) PBWTGPU_MSP_##a, __VA_ARGS__)
#define PBWTGPU_MSP_p                     matchSequencesSweepSparse (p
#define PBWTGPU_MSP_PBWT                  matchSequencesSweepSparse_cpu (PBWT

#include "pbwtMatch.c"		/* found through -I<reference> */

#undef matchMaximalWithin
#undef matchLongWithin2
#undef matchSequencesSweep
#undef matchSequencesSweepSparse

#define PBWT_GPU_UNITY 1	/* same TU as pbwtMatch.c: its file-static matchLengthHist (-stats) is in reach */
#include "pbwtGpu.c"
