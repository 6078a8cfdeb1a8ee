// PhmmPreprocessor.hpp -- P7HmmList -> one flattened int8 table for the device.
// Same class name, constructor and getters as host/phmm/PhmmPreprocessor.hpp:15-38.
#ifndef HAVAC_PHMM_PREPROCESSOR_HPP
#define HAVAC_PHMM_PREPROCESSOR_HPP

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "p7HmmReader.h"

class PhmmPreprocessor {
public:
    // Projects every model for `desiredPvalue` and lays the tables back to back, no separator
    // (host/phmm/PhmmPreprocessor.cpp:9-31): diagonals run across model boundaries.
    PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue = 0.05f);
    // Boundary mode (not in the reference): two rows of -128 follow every model, so a
    // diagonal cannot carry a score from one model into the next (255 - 128 - 128 < 0).  getModelStarts() gives the
    // global row of each model's first position.
    PhmmPreprocessor(P7HmmList *phmmList, const float desiredPvalue, bool boundaryMode);
    const std::vector<uint32_t> &getModelStarts() const { return modelStarts_; }
    std::shared_ptr<std::vector<int8_t>> getProcessedPhmmData() { return data_; }
    uint32_t getPhmmLengthInBytes() const { return (uint32_t)data_->size(); }
    uint32_t getPhmmListLengthInVectors() const { return rows_; }

private:
    void projectAll(P7HmmList *phmmList, const float desiredPvalue, const std::vector<size_t> &byteOffsets);
    std::shared_ptr<std::vector<int8_t>> data_;
    uint32_t rows_ = 0;
    std::vector<uint32_t> modelStarts_;
};
#endif
