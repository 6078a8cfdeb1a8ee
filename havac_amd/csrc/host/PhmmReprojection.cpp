// PhmmReprojection.cpp -- see PhmmReprojection.h.  Compile with -ffp-contract=off.
//
// Every intermediate below keeps the width (float or double) the reference
// gives it, because the int8 result after rounding depends on it; the widths
// are called out line by line.  Citations: /root/reference/PhmmReprojection/PhmmReprojection.cpp.
#include "PhmmReprojection.h"

#include <cmath>
#include <cstddef>

namespace {
constexpr double kGumbelSmallP = 5e-9;                 // :6
constexpr double kLn2 = 0.69314718055994529;           // :7
constexpr float kLog2e = 1.44269504089;                // :95,113 (double literal narrowed to float)

inline float saturateToInt8Range(float v) {            // :99-105, :137-142
    if (v < -128) v = -128;
    if (v > 127) v = 127;
    return v;
}
}  // namespace

extern "C" {

double esl_gumbel_invsurv(double p, double mu, double lambda) {
    // :28-31  for tiny p, ln(-ln(1-p)) ~ ln p ~ (p^p - 1)/p
    const double inner = (p < kGumbelSmallP) ? (std::pow(p, p) - 1.0) / p : std::log(-1.0 * std::log(1.0 - p));
    return mu - inner / lambda;
}

float findThreshold256ScalingFactor(const P7Hmm *phmm, float pValue) {
    const float mu = phmm->stats.msvGumbelMu;                        // :37
    const float lambda = phmm->stats.msvGumbelLambda;                // :38
    const float maxL = phmm->header.maxLength;                       // :39 uint32 -> float
    const float L = phmm->header.modelLength;                        // :40 uint32 -> float

    const double fullModelScore = esl_gumbel_invsurv(pValue, mu, lambda);   // :43 double

    // single-hit model corrections, all float with logf (:45-51)
    const float nLoop = std::log(maxL / (maxL + 3));                 // float overload == logf
    const float nLoopTotal = nLoop * maxL;
    const float nEscape = std::log(3.0f / (maxL + 3));
    const float bToMk = std::log(2.0f / (L * (L + 1)));
    const float eToC = std::log(1.0f / 2);
    const float core = (nEscape + nLoopTotal + nEscape + bToMk + eToC);     // :50-51 float, left to right

    // background (:53-56): first log on a float (float overload), second on a double expression
    const float bgLoopP = maxL / (maxL + 1);
    const float bgLoopTotal = maxL * std::log(bgLoopP);
    const float bgMove = std::log(1.0 - bgLoopP);                    // double log, narrowed
    const float bg = bgLoopTotal + bgMove;

    const float thresholdNats = (fullModelScore * kLn2) + bg - core; // :58-59 evaluated in double, narrowed
    const float thresholdBits = thresholdNats / kLn2;                // :61 double division, narrowed
    return 256.0f / thresholdBits;                                   // :63
}

float emissionScoreToProjectedScore(float emissionScore, float scoreMultiplier) {
    // :96  y = -log2(e) * (s - 2/log2(e)) * m, all float
    float y = -kLog2e * (emissionScore - (2 / kLog2e)) * scoreMultiplier;
    y = std::round(y);                                               // :97 half away from zero
    return saturateToInt8Range(y);
}

void p7HmmProjectForThreshold256(const P7Hmm *phmm, float desiredPValue, int8_t *outputArray) {
    const float L = phmm->header.modelLength;                        // :110
    const float m = findThreshold256ScalingFactor(phmm, desiredPValue);
    const float alpha = 2 * m;                                       // :114
    const float beta = kLog2e * m;                                   // :115
    const uint32_t K = p7HmmGetAlphabetCardinality(phmm);            // :117
    const float *scores = phmm->model.matchEmissionScores;
    for (size_t i = 0; i < K * L; i++) {                             // :118 (float bound, as in the reference)
        float y = alpha - (scores[i] * beta);                        // :133 float multiply then float subtract, no FMA
        y = std::round(y);                                           // :134
        outputArray[i] = (int8_t)saturateToInt8Range(y);             // :137-143
    }
}

}  // extern "C"
