// PhmmPreprocessor.hpp -- P7HmmList -> one flattened int8 table for the device.
// Same class name, constructor and getters as host/phmm/PhmmPreprocessor.hpp:15-38.
#ifndef HAVAC_PHMM_PREPROCESSOR_HPP
#define HAVAC_PHMM_PREPROCESSOR_HPP

#include <cstdint>
#include <memory>
#include <vector>

#include "p7HmmReader.h"

class PhmmPreprocessor {
public:
    // Projects every model for `desiredPvalue` and lays the tables back to back, no separator
    // (host/phmm/PhmmPreprocessor.cpp:9-31): diagonals run across model boundaries.
    PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue = 0.05f);
    std::shared_ptr<std::vector<int8_t>> getProcessedPhmmData() { return data_; }
    uint32_t getPhmmLengthInBytes() const { return (uint32_t)data_->size(); }
    uint32_t getPhmmListLengthInVectors() const { return rows_; }

private:
    std::shared_ptr<std::vector<int8_t>> data_;
    uint32_t rows_ = 0;
};
#endif
