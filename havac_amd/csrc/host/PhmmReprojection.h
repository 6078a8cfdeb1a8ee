// PhmmReprojection.h -- p-value -> int8 score projection (host CPU, by design).
//
// Same three entry points, names and argument meaning as the reference's
// PhmmReprojection/PhmmReprojection.h:14-31.  The projection is part of the
// reference's HOST path (it runs once per model file on the CPU there too,
// host/phmm/PhmmPreprocessor.cpp:22-24) and must stay on the CPU with the
// reference's float/double mix to give the same int8 tables.
// Must be compiled with -ffp-contract=off: the reference build has no FMA.
//
// Parity status: UNPINNED against a reference build.  The reference file needs
// the un-vendored <p7HmmReader.h>; writing a stand-in header to compile it is
// not allowed here, and the reference ships no known-answer vectors for it.
// The restatement is pinned by our own tests only (worked example of SURVEY.md
// A.5, monotonicity, saturation, rounding-mode cases).
#ifndef HAVAC_PHMM_REPROJECTION_H
#define HAVAC_PHMM_REPROJECTION_H

#include <stdint.h>

#include "p7HmmReader.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Inverse survival function of a Gumbel(mu, lambda): the score whose tail
 * probability is p.  PhmmReprojection.cpp:15-31 (adapted there from Easel). */
double esl_gumbel_invsurv(double p, double mu, double lambda);

/* Multiplier that maps bit scores so that a diagonal reaching p-value
 * `pValue` accumulates exactly 256.  PhmmReprojection.cpp:36-64. */
float findThreshold256ScalingFactor(const struct P7Hmm *phmm, float pValue);

/* One file value (-ln p) to its rounded, saturated projected score, returned
 * as a float.  PhmmReprojection.cpp:90-107. */
float emissionScoreToProjectedScore(float emissionScore, float scoreMultiplier);

/* All L*K match scores of one model into int8.  PhmmReprojection.cpp:109-145. */
void p7HmmProjectForThreshold256(const struct P7Hmm *phmm, float desiredPValue, int8_t *outputArray);

#ifdef __cplusplus
}
#endif
#endif
