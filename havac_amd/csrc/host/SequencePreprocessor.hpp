// SequencePreprocessor.hpp -- FASTA residues -> HAVAC's 2-bit packed, segment-padded sequence.
// Same class name, constructor and getters as host/sequence/SequencePreprocessor.hpp:18-57.
#ifndef HAVAC_SEQUENCE_PREPROCESSOR_HPP
#define HAVAC_SEQUENCE_PREPROCESSOR_HPP

#include <cstdint>
#include <vector>

#include "FastaVector.h"

class SequencePreprocessor {
public:
    // Packs every character of fastaVector->sequence (record terminators included) and pads
    // with zero bytes to a whole number of 12288-symbol segments.
    explicit SequencePreprocessor(struct FastaVector *fastaVector);
    std::vector<uint8_t> &getCompressedSequenceBuffer() { return packed_; }
    uint32_t getCompressedSequenceLengthInSegments() const { return segments_; }
    uint32_t getCompressedSequenceLengthInSymbols() const { return symbols_; }
    uint32_t getCompressedSquenceLengthInBytes() const { return bytes_; }   // (sic) the reference's spelling

    // One character to its 2-bit code (may return 4 for 'Y'/'y': the reference's precedence slip).
    static uint8_t getCompressedSymbol(char originalSymbol);

private:
    uint32_t originalLength_ = 0, segments_ = 0, symbols_ = 0, bytes_ = 0;
    std::vector<uint8_t> packed_;
};
#endif
