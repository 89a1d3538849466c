// pbwt_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels for the PBWT hot path.
//
// HBM layout:
//   ring slot s:  A[s][Mpad] int32  — prefix array a_k; bits 31/30 carry the alleles of the haplotype at
//                                     that position at the slot's site / the following site ("tags"), a < 2^30
//                 D[s][Mpad+64]     — divergence d_k[0..M] (start-position form, pbwt.h:82, with sentinels)
//   tile summaries                  — per tile of T positions of the NEXT launch's input order, built with
//                                     commutative atomics by the launch that scatters into that order:
//                                     single-site steps: int4 {cnt0, last0+1, last1+1, maxd};
//                                     two-site steps: 3 int4 {c[4]}, {last[4]+1}, {maxd} over the 2-bit keys
//   bit columns  C[k][wpc] uint32   — original order (gather by a) or sorted order (index by position)
//
// Kernels: prepare/prepare2 (tags + summaries of the first site(s) of a pass), step2 (two sites of
// pbwtCursorForwardsA/AD per launch, build side), step1/step (one site per launch: read side, large M),
// and the batch consumers (checksum, maxWithin / longWithin sweeps pbwtMatch.c:85-142, pack3
// encode/decode pbwtCore.c:240-305, query sweep pbwtMatch.c:363-443).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pbwt_k_common.h"
#include "pbwt_k_step2.h"
#include "pbwt_k_chain.h"
#include "pbwt_k_fill.h"
#include "pbwt_k_fillseq.h"
#include "pbwt_k_step1.h"
#include "pbwt_k_misc.h"
#include "pbwt_k_sweep.h"
#include "pbwt_k_codec.h"
#include "pbwt_k_query.h"
