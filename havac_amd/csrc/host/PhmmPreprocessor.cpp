// PhmmPreprocessor.cpp -- restates host/phmm/PhmmPreprocessor.cpp:9-31.
#include "PhmmPreprocessor.hpp"

#include <stdexcept>
#include <string>

#include "HostThreads.hpp"
#include "PhmmReprojection.h"

// The tables are [row][A,C,G,T] (host/phmm/PhmmPreprocessor.cpp:15 allocates rows * 4), but the projection writes
// cardinality * rows scores (PhmmReprojection.cpp:117-118): an `ALPH amino` model (20 per row) would run past the
// buffer.  The reference has the same flaw and no check; here such a file is an error.
static void requireNucleotideModels(const P7HmmList *phmmList) {
    for (uint32_t i = 0; i < phmmList->count; i++)
        if (p7HmmGetAlphabetCardinality(&phmmList->phmms[i]) != 4)
            throw std::runtime_error("model " + std::to_string(i) + " of the phmm file is not a nucleotide model "
                                     "(only nucleotide phmms with 4 scores/position are supported).");
}

PhmmPreprocessor::PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue) {
    requireNucleotideModels(phmmList);
    for (uint32_t i = 0; i < phmmList->count; i++) rows_ += phmmList->phmms[i].header.modelLength;
    data_ = std::make_shared<std::vector<int8_t>>((size_t)rows_ * 4);
    std::vector<size_t> at(phmmList->count, 0);
    for (uint32_t i = 1; i < phmmList->count; i++) at[i] = at[i - 1] + (size_t)phmmList->phmms[i - 1].header.modelLength * 4;
    projectAll(phmmList, desiredPvalue, at);
}

// every model is projected on its own (host/phmm/PhmmPreprocessor.cpp:21-30): side by side on the host's cores
void PhmmPreprocessor::projectAll(P7HmmList *phmmList, const float desiredPvalue, const std::vector<size_t> &byteOffsets) {
    int8_t *const out = data_->data();
    const unsigned threads = rows_ < 4096 ? 1u : havacHostThreads(phmmList->count);
    havacParallelFor(phmmList->count, threads, [&](size_t i) {
        p7HmmProjectForThreshold256(&phmmList->phmms[i], desiredPvalue, out + byteOffsets[i]);
    });
}

PhmmPreprocessor::PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue, bool boundaryMode) {
    requireNucleotideModels(phmmList);
    const uint32_t gap = boundaryMode ? 2u : 0u;
    for (uint32_t i = 0; i < phmmList->count; i++) {
        modelStarts_.push_back(rows_);
        rows_ += phmmList->phmms[i].header.modelLength + gap;
    }
    data_ = std::make_shared<std::vector<int8_t>>((size_t)rows_ * 4, (int8_t)-128);   // separator rows stay -128
    std::vector<size_t> at(phmmList->count, 0);
    for (uint32_t i = 0; i < phmmList->count; i++) at[i] = (size_t)modelStarts_[i] * 4;
    projectAll(phmmList, desiredPvalue, at);
}
