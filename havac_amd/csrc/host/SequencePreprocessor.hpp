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
    // Boundary mode (not in the reference): every record keeps its own columns -- its
    // residues and its terminator column, as the reference counts them -- followed, at an even column, by one
    // separator pair that the device scores -128 on every row, so no diagonal survives from one record into the
    // next.  Characters other than a/c/g (terminator, N, ambiguity codes) are 'T', as in the reference's per-pair
    // CPU SSV (host/test/Ssv.cpp:29-34); rand() is not used.
    SequencePreprocessor(struct FastaVector *fastaVector, bool boundaryMode);
    // Both strands (not in the reference, which was benchmarked with nhmmer --watson):
    // doubles the buffer.  The second half repeats the first with every record's residues reverse-complemented in
    // place (terminator, separator and padding columns are copied), so a hit at column Nf + c is a hit on the
    // reverse strand of the record that owns column c.  `starts`/`residues` give each record's first column and
    // residue count in the forward layout.  Returns Nf, the forward half's padded symbol count.
    uint64_t appendReverseStrand(const std::vector<uint64_t> &starts, const std::vector<uint64_t> &residues);
    // boundary mode only: one bit per aligned symbol pair, set on separator pairs; global column of each record
    std::vector<uint8_t> &getSeparatorMask() { return mask_; }
    const std::vector<uint64_t> &getRecordStarts() const { return recordStarts_; }
    const std::vector<uint64_t> &getRecordLengths() const { return recordLengths_; }   // terminator column included
    std::vector<uint8_t> &getCompressedSequenceBuffer() { return packed_; }
    uint32_t getCompressedSequenceLengthInSegments() const { return segments_; }
    uint32_t getCompressedSequenceLengthInSymbols() const { return symbols_; }
    uint32_t getCompressedSquenceLengthInBytes() const { return bytes_; }   // (sic) the reference's spelling

    // One character to its 2-bit code (may return 4 for 'Y'/'y': the reference's precedence slip).
    static uint8_t getCompressedSymbol(char originalSymbol);

    // For packing on the GPU (include/havac_dev.h: havac_dev_write_sequence_chars): the
    // columns that are not a/c/g/t and the symbol each of them gets -- getCompressedSymbol called for exactly those
    // characters, in order, so rand() is drawn as often and in the same order as by the constructor above, and the
    // device buffer equals getCompressedSequenceBuffer() byte for byte (the 'Y' spill included: a final 'Y' that
    // draws 1 sets the low bit of the first padding column).
    static void collectPatches(const struct FastaVector *fastaVector, std::vector<uint64_t> &columns,
                               std::vector<uint8_t> &symbols);

private:
    uint32_t originalLength_ = 0, segments_ = 0, symbols_ = 0, bytes_ = 0;
    std::vector<uint8_t> packed_;
    std::vector<uint8_t> mask_;
    std::vector<uint64_t> recordStarts_, recordLengths_;
};
#endif
