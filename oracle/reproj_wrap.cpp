/*
 * reproj_wrap.cpp -- C entry points around the REFERENCE's int8 projection.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; the functions it calls are the reference's own
 * PhmmReprojection/PhmmReprojection.cpp, compiled from where it lies under /root/reference by oracle/Makefile (target
 * `_ref`) into oracle/_ref/libreprojection_ref.so.  No reference source is copied into this repository.
 *
 * What this build is and is not.  PhmmReprojection.cpp includes "p7HmmReader.h" of the un-vendored P7HmmReader
 * submodule (empty in the reference tree).  It is compiled here against the PRODUCT's own reader header,
 * havac_amd/csrc/host/p7HmmReader.h (the header the product ships anyway, with the field
 * names the reference uses), not against the reference's.  The arithmetic -- every line of esl_gumbel_invsurv,
 * findThreshold256ScalingFactor, emissionScoreToProjectedScore and p7HmmProjectForThreshold256 -- is the reference's;
 * the struct it reads its five inputs from is ours.  This is therefore a DIFFERENTIAL check with a substituted header,
 * used to capture the known-answer vectors of SURVEY.md section 8c G7 (tests/golden/g7_projection.npz); it is not
 * "the reference compiled here" in the sense the softSsv object is.
 */
#include <cstdint>
#include <cstring>

#include "PhmmReprojection.h" /* the reference's, from -I$(REFERENCE)/PhmmReprojection; pulls in our p7HmmReader.h */

double esl_gumbel_invsurv(double p, double mu, double lambda); /* defined in the reference file, not in its header */

static P7Hmm make(float mu, float lambda, uint32_t maxLength, uint32_t modelLength, const float *emissions) {
    P7Hmm h;
    std::memset(&h, 0, sizeof h);
    h.header.alphabet = P7HmmReaderAlphabetDna;
    h.header.maxLength = maxLength;
    h.header.modelLength = modelLength;
    h.stats.msvGumbelMu = mu;
    h.stats.msvGumbelLambda = lambda;
    h.model.matchEmissionScores = const_cast<float *>(emissions);
    return h;
}

extern "C" {

double reproj_ref_invsurv(double p, double mu, double lambda) { return esl_gumbel_invsurv(p, mu, lambda); }

float reproj_ref_scale(float mu, float lambda, uint32_t maxLength, uint32_t modelLength, float p) {
    const P7Hmm h = make(mu, lambda, maxLength, modelLength, nullptr);
    return findThreshold256ScalingFactor(&h, p);
}

float reproj_ref_score(float emissionScore, float multiplier) { return emissionScoreToProjectedScore(emissionScore, multiplier); }

/* emissions: modelLength x 4 file values (-ln p, +inf for '*'); out: modelLength x 4 int8 */
void reproj_ref_project(float mu, float lambda, uint32_t maxLength, uint32_t modelLength, float p, const float *emissions,
                        int8_t *out) {
    const P7Hmm h = make(mu, lambda, maxLength, modelLength, emissions);
    p7HmmProjectForThreshold256(&h, p, out);
}

}
