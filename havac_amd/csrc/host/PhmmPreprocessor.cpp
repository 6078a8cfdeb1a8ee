// PhmmPreprocessor.cpp -- restates host/phmm/PhmmPreprocessor.cpp:9-31.
#include "PhmmPreprocessor.hpp"

#include "PhmmReprojection.h"

PhmmPreprocessor::PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue) {
    for (uint32_t i = 0; i < phmmList->count; i++) rows_ += phmmList->phmms[i].header.modelLength;
    data_ = std::make_shared<std::vector<int8_t>>((size_t)rows_ * 4);
    size_t at = 0;
    for (uint32_t i = 0; i < phmmList->count; i++) {
        p7HmmProjectForThreshold256(&phmmList->phmms[i], desiredPvalue, data_->data() + at);
        at += (size_t)phmmList->phmms[i].header.modelLength * 4;
    }
}

PhmmPreprocessor::PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue, bool boundaryMode) {
    const uint32_t gap = boundaryMode ? 2u : 0u;
    for (uint32_t i = 0; i < phmmList->count; i++) {
        modelStarts_.push_back(rows_);
        rows_ += phmmList->phmms[i].header.modelLength + gap;
    }
    data_ = std::make_shared<std::vector<int8_t>>((size_t)rows_ * 4, (int8_t)-128);   // separator rows stay -128
    for (uint32_t i = 0; i < phmmList->count; i++)
        p7HmmProjectForThreshold256(&phmmList->phmms[i], desiredPvalue, data_->data() + (size_t)modelStarts_[i] * 4);
}
