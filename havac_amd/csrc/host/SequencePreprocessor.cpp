// SequencePreprocessor.cpp -- restates host/sequence/SequencePreprocessor.cpp:9-84.
//
// Quirks kept on purpose, because they decide which symbol lands in which column:
//  * the length is sequence.count INCLUDING the '\0' after every record (:12-13); each '\0', 'N'
//    or other unknown character becomes a random nucleotide;
//  * for any non-ACGT character rand()%2 is drawn first (:74); two-fold ambiguity codes use that
//    draw, everything else draws rand()%4 afterwards (:82) -- the rand() call ORDER is kept, so a
//    caller that seeds with srand(s) gets the reference's sequence for the same libc;
//  * 'Y' returns randMod2 << 1 + 1, which C parses as randMod2 << 2, i.e. 0 or 4 (:77).  The value 4
//    sets a bit of the NEXT symbol's field, which that symbol's own masked write then clears unless
//    'Y' is the very last character;
//  * padding bytes are zero, i.e. symbol 0 = 'A' (:41).
#include "SequencePreprocessor.hpp"

#include <algorithm>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {
constexpr uint32_t kSegment = 12288;   // NUM_CELL_PROCESSORS, device/PublicDefines.h:18-22
}

namespace {
// 0..3 for acgtACGT, 0xff for every character that draws from rand()
struct CodeTable {
    uint8_t code[256];
    CodeTable() {
        for (int i = 0; i < 256; i++) code[i] = 0xff;
        code[(int)'a'] = code[(int)'A'] = 0;
        code[(int)'c'] = code[(int)'C'] = 1;
        code[(int)'g'] = code[(int)'G'] = 2;
        code[(int)'t'] = code[(int)'T'] = 3;
    }
};
const CodeTable kCodes;
}  // namespace

SequencePreprocessor::SequencePreprocessor(FastaVector *fastaVector) {
    originalLength_ = (uint32_t)fastaVector->sequence.count;
    segments_ = (originalLength_ + (kSegment - 1)) / kSegment;
    symbols_ = segments_ * kSegment;
    bytes_ = symbols_ / 4;
    packed_.assign(bytes_, 0);
    const unsigned char *chars = reinterpret_cast<const unsigned char *>(fastaVector->sequence.charData);
    const uint32_t count = (uint32_t)fastaVector->sequence.count;
    // Same result as the reference's one-character-at-a-time loop (host/sequence/SequencePreprocessor.cpp:37-59),
    // produced a byte (four characters) at a time when all four are plain nucleotides; any other character goes
    // through the reference-shaped path, in order, so rand() is called exactly as often and in the same order.
    auto slow = [&](uint32_t i) {
        const uint8_t code = getCompressedSymbol((char)chars[i]);
        const uint32_t byte = i / 4;
        const uint8_t shift = (uint8_t)((i % 4) * 2);
        packed_[byte] &= (uint8_t)~(uint8_t)(0x3u << shift);     // clear this symbol's field (:52-54)
        packed_[byte] |= (uint8_t)(code << shift);               // the unmasked OR of :56-57 ('Y' may spill upward)
    };
    uint32_t i = 0;
    for (; i + 4 <= count; i += 4) {
        const uint8_t a = kCodes.code[chars[i]], b = kCodes.code[chars[i + 1]], c = kCodes.code[chars[i + 2]],
                      d = kCodes.code[chars[i + 3]];
        if ((a | b | c | d) <= 3) {
            packed_[i / 4] = (uint8_t)(a | (b << 2) | (c << 4) | (d << 6));
        } else {
            slow(i); slow(i + 1); slow(i + 2); slow(i + 3);
        }
    }
    for (; i < count; i++) slow(i);
}

SequencePreprocessor::SequencePreprocessor(FastaVector *fastaVector, bool boundaryMode) {
    if (!boundaryMode) { *this = SequencePreprocessor(fastaVector); return; }
    // layout pass: record j occupies [start_j, start_j + len_j), then padding to an even column, then 2 separators
    uint64_t at = 0;
    for (size_t j = 0; j < fastaVector->metadata.count; j++) {
        const uint64_t begin = j ? fastaVector->metadata.data[j - 1].sequenceEndPosition : 0;
        const uint64_t len = fastaVector->metadata.data[j].sequenceEndPosition - begin;     // terminator included
        recordStarts_.push_back(at);
        recordLengths_.push_back(len);
        at += len;
        at += at & 1;        // separator pairs sit on even columns
        at += 2;
    }
    originalLength_ = (uint32_t)at;
    segments_ = (uint32_t)((at + (kSegment - 1)) / kSegment);
    symbols_ = segments_ * kSegment;
    bytes_ = symbols_ / 4;
    packed_.assign(bytes_, 0);
    mask_.assign(symbols_ / 16, 0);
    auto put = [&](uint64_t col, uint8_t code) { packed_[col / 4] |= (uint8_t)(code << ((col % 4) * 2)); };
    const unsigned char *chars = reinterpret_cast<const unsigned char *>(fastaVector->sequence.charData);
    for (size_t j = 0; j < recordStarts_.size(); j++) {
        const uint64_t begin = j ? fastaVector->metadata.data[j - 1].sequenceEndPosition : 0;
        for (uint64_t i = 0; i < recordLengths_[j]; i++) {
            uint8_t code = kCodes.code[chars[begin + i]];
            if (code > 2) code = 3;                                  // host/test/Ssv.cpp:29-34: default -> 3
            put(recordStarts_[j] + i, code);
        }
        uint64_t sep = recordStarts_[j] + recordLengths_[j];
        sep += sep & 1;
        mask_[sep / 16] |= (uint8_t)(1u << ((sep / 2) % 8));
    }
    // the padding after the last record is masked too: nothing can hit there
    for (uint64_t col = at; col < symbols_; col += 2) mask_[col / 16] |= (uint8_t)(1u << ((col / 2) % 8));
}

uint64_t SequencePreprocessor::appendReverseStrand(const std::vector<uint64_t> &starts,
                                                   const std::vector<uint64_t> &residues) {
    const uint64_t nf = symbols_;
    const size_t fbytes = packed_.size();
    packed_.resize(2 * fbytes);
    std::copy(packed_.begin(), packed_.begin() + fbytes, packed_.begin() + fbytes);      // terminators, padding, ...
    auto get = [&](uint64_t col) -> uint8_t { return (uint8_t)((packed_[col / 4] >> ((col % 4) * 2)) & 3u); };
    auto set = [&](uint64_t col, uint8_t code) {
        uint8_t &b = packed_[col / 4];
        b = (uint8_t)((b & ~(0x3u << ((col % 4) * 2))) | (code << ((col % 4) * 2)));
    };
    for (size_t j = 0; j < starts.size(); j++)
        for (uint64_t i = 0; i < residues[j]; i++)
            set(nf + starts[j] + i, (uint8_t)(3u - get(starts[j] + residues[j] - 1 - i)));   // A<->T, C<->G
    if (!mask_.empty()) {
        const size_t mbytes = mask_.size();
        mask_.resize(2 * mbytes);
        std::copy(mask_.begin(), mask_.begin() + mbytes, mask_.begin() + mbytes);
    }
    symbols_ *= 2; segments_ *= 2; bytes_ *= 2;
    return nf;
}

namespace {
// index of the first character at or after `from` that is not a/c/g/t (either case), or `count`
#if defined(__x86_64__)
__attribute__((target("avx2")))
uint64_t nextOtherAvx2(const unsigned char *chars, uint64_t from, uint64_t count) {
    const __m256i fold = _mm256_set1_epi8(0x20);
    const __m256i a = _mm256_set1_epi8('a'), c = _mm256_set1_epi8('c'), g = _mm256_set1_epi8('g'), t = _mm256_set1_epi8('t');
    uint64_t i = from;
    for (; i + 32 <= count; i += 32) {
        const __m256i x = _mm256_or_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(chars + i)), fold);
        const __m256i plain = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, a), _mm256_cmpeq_epi8(x, c)),
                                              _mm256_or_si256(_mm256_cmpeq_epi8(x, g), _mm256_cmpeq_epi8(x, t)));
        const uint32_t other = ~(uint32_t)_mm256_movemask_epi8(plain);   // x | 0x20 == 'a' only for 'A' and 'a', etc.
        if (other) return i + (uint64_t)__builtin_ctz(other);
    }
    for (; i < count; i++)
        if (kCodes.code[chars[i]] > 3) return i;
    return count;
}
#endif
uint64_t nextOther(const unsigned char *chars, uint64_t from, uint64_t count) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return nextOtherAvx2(chars, from, count);
#endif
    for (uint64_t i = from; i < count; i++)
        if (kCodes.code[chars[i]] > 3) return i;
    return count;
}
}  // namespace

void SequencePreprocessor::collectPatches(const FastaVector *fastaVector, std::vector<uint64_t> &columns,
                                          std::vector<uint8_t> &symbols) {
    const unsigned char *chars = reinterpret_cast<const unsigned char *>(fastaVector->sequence.charData);
    const uint64_t count = fastaVector->sequence.count;
    columns.clear();
    symbols.clear();
    for (uint64_t i = nextOther(chars, 0, count); i < count; i = nextOther(chars, i + 1, count)) {
        const uint8_t code = getCompressedSymbol((char)chars[i]);
        columns.push_back(i);
        symbols.push_back((uint8_t)(code & 3u));
        // 'Y' drew 1: the reference ORs 4 << shift into the byte, i.e. sets the low bit of the next symbol of the same
        // byte; every later symbol clears its own field before writing, so the bit survives only behind the last character
        if (code == 4 && i + 1 == count && (i % 4) != 3) {
            columns.push_back(i + 1);
            symbols.push_back(1);
        }
    }
}

uint8_t SequencePreprocessor::getCompressedSymbol(const char c) {
    switch (c) {
        case 'a': case 'A': return 0;
        case 'c': case 'C': return 1;
        case 'g': case 'G': return 2;
        case 't': case 'T': return 3;
        default: break;
    }
    const uint8_t coin = (uint8_t)(std::rand() % 2);             // drawn for EVERY other character (:74)
    switch (c) {
        case 'r': case 'R': return (uint8_t)(coin * 2);          // A or G
        case 'y': case 'Y': return (uint8_t)(coin << 2);         // reference: coin << 1 + 1  ==  coin << 2
        case 's': case 'S': return (uint8_t)(coin + 1);          // C or G
        case 'w': case 'W': return (uint8_t)(coin * 3);          // A or T
        case 'k': case 'K': return (uint8_t)(coin + 2);          // G or T
        case 'm': case 'M': return coin;                         // A or C
        default: return (uint8_t)(std::rand() % 4);              // '\0', N, 3-fold codes, anything else (:82)
    }
}
