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
