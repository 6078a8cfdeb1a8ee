// FastaVector.cpp -- see FastaVector.h.  Own code; the reference's dependency is not vendored.
#include "FastaVector.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

bool reserve(FastaVectorString &s, size_t extra) {
    if (s.count + extra <= s.capacity) return true;
    size_t cap = s.capacity ? s.capacity : 4096;
    while (cap < s.count + extra) cap *= 2;
    char *p = static_cast<char *>(std::realloc(s.charData, cap));
    if (!p) return false;
    s.charData = p;
    s.capacity = cap;
    return true;
}

bool push(FastaVectorString &s, const char *src, size_t n) {
    if (!reserve(s, n)) return false;
    std::memcpy(s.charData + s.count, src, n);
    s.count += n;
    return true;
}

bool pushMeta(FastaVectorMetadataVector &m, FastaVectorMetadata v) {
    if (m.count == m.capacity) {
        size_t cap = m.capacity ? m.capacity * 2 : 64;
        auto *p = static_cast<FastaVectorMetadata *>(std::realloc(m.data, cap * sizeof(FastaVectorMetadata)));
        if (!p) return false;
        m.data = p;
        m.capacity = cap;
    }
    m.data[m.count++] = v;
    return true;
}

}  // namespace

extern "C" {

FastaVectorReturnCode fastaVectorInit(FastaVector *fv) {
    std::memset(fv, 0, sizeof(*fv));
    if (!reserve(fv->sequence, 1) || !reserve(fv->header, 1)) return FASTA_VECTOR_ALLOCATION_FAIL;
    return FASTA_VECTOR_OK;
}

void fastaVectorDealloc(FastaVector *fv) {
    if (!fv) return;
    std::free(fv->sequence.charData);
    std::free(fv->header.charData);
    std::free(fv->metadata.data);
    std::memset(fv, 0, sizeof(*fv));
}

FastaVectorReturnCode fastaVectorAddSequenceToList(FastaVector *fv, const char *header, size_t headerLength,
                                                   const char *sequence, size_t sequenceLength) {
    const char zero = '\0';
    if (!push(fv->header, header, headerLength) || !push(fv->header, &zero, 1)) return FASTA_VECTOR_ALLOCATION_FAIL;
    if (!push(fv->sequence, sequence, sequenceLength) || !push(fv->sequence, &zero, 1)) return FASTA_VECTOR_ALLOCATION_FAIL;
    if (!pushMeta(fv->metadata, {fv->header.count, fv->sequence.count})) return FASTA_VECTOR_ALLOCATION_FAIL;
    return FASTA_VECTOR_OK;
}

FastaVectorReturnCode fastaVectorReadFasta(const char *path, FastaVector *fv) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return FASTA_VECTOR_FILE_OPEN_FAIL;
    const char zero = '\0';
    bool inRecord = false, inHeader = false, lineStart = true;
    static const size_t kBuf = 1 << 20;
    char *buf = static_cast<char *>(std::malloc(kBuf));
    if (!buf) { std::fclose(f); return FASTA_VECTOR_ALLOCATION_FAIL; }
    FastaVectorReturnCode rc = FASTA_VECTOR_OK;
    auto closeRecord = [&]() -> bool {
        if (!inRecord) return true;
        inRecord = false;
        return push(fv->sequence, &zero, 1) && pushMeta(fv->metadata, {fv->header.count, fv->sequence.count});
    };
    // Works a line segment at a time: memchr finds the end of the line inside the buffer, residues are
    // appended in bulk and blanks are squeezed out afterwards only if the segment had any.
    size_t got;
    while (rc == FASTA_VECTOR_OK && (got = std::fread(buf, 1, kBuf, f)) > 0) {
        size_t i = 0;
        while (i < got && rc == FASTA_VECTOR_OK) {
            const char *nl = static_cast<const char *>(std::memchr(buf + i, '\n', got - i));
            const size_t end = nl ? (size_t)(nl - buf) : got;       // segment [i, end), newline (if any) at end
            if (inHeader) {
                size_t n = end - i;
                if (!push(fv->header, buf + i, n)) { rc = FASTA_VECTOR_ALLOCATION_FAIL; break; }
                // drop carriage returns from what was just appended
                size_t w = fv->header.count - n;
                for (size_t r = w; r < fv->header.count; r++)
                    if (fv->header.charData[r] != '\r') fv->header.charData[w++] = fv->header.charData[r];
                fv->header.count = w;
                if (nl) {
                    inHeader = false;
                    lineStart = true;
                    if (!push(fv->header, &zero, 1)) rc = FASTA_VECTOR_ALLOCATION_FAIL;
                }
            } else if (i < end) {
                if (lineStart && buf[i] == '>') {
                    if (!closeRecord()) { rc = FASTA_VECTOR_ALLOCATION_FAIL; break; }
                    inRecord = true;
                    inHeader = true;
                    lineStart = false;
                    i++;                      // past '>' and handle the rest of the segment as header text
                    continue;
                }
                lineStart = false;
                size_t n = end - i;
                if (!inRecord) {              // residues before any header: an unnamed record
                    bool any = false;
                    for (size_t r = i; r < end && !any; r++) any = !(buf[r] == '\r' || buf[r] == ' ' || buf[r] == '\t');
                    if (any) {
                        inRecord = true;
                        if (!push(fv->header, &zero, 1)) { rc = FASTA_VECTOR_ALLOCATION_FAIL; break; }
                    }
                }
                if (inRecord) {
                    if (!push(fv->sequence, buf + i, n)) { rc = FASTA_VECTOR_ALLOCATION_FAIL; break; }
                    const char *seg = fv->sequence.charData + fv->sequence.count - n;
                    if (std::memchr(seg, '\r', n) || std::memchr(seg, ' ', n) || std::memchr(seg, '\t', n)) {
                        size_t w = fv->sequence.count - n;
                        for (size_t r = w; r < fv->sequence.count; r++) {
                            const char c = fv->sequence.charData[r];
                            if (c != '\r' && c != ' ' && c != '\t') fv->sequence.charData[w++] = c;
                        }
                        fv->sequence.count = w;
                    }
                }
                if (nl) lineStart = true;
            } else if (nl) {
                lineStart = true;             // empty line
            }
            i = nl ? end + 1 : got;
        }
    }
    if (rc == FASTA_VECTOR_OK && std::ferror(f)) rc = FASTA_VECTOR_FILE_READ_FAIL;
    if (rc == FASTA_VECTOR_OK && inHeader && !push(fv->header, &zero, 1)) rc = FASTA_VECTOR_ALLOCATION_FAIL;
    if (rc == FASTA_VECTOR_OK && !closeRecord()) rc = FASTA_VECTOR_ALLOCATION_FAIL;
    std::free(buf);
    std::fclose(f);
    return rc;
}

bool fastaVectorGetLocalSequencePositionFromGlobal(const FastaVector *fv, size_t globalPosition,
                                                   FastaVectorLocalPosition *out) {
    // binary search for the first record whose end lies beyond the position
    size_t lo = 0, hi = fv->metadata.count;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (fv->metadata.data[mid].sequenceEndPosition <= globalPosition) lo = mid + 1; else hi = mid;
    }
    if (lo == fv->metadata.count) return false;   // beyond the last record: padding
    size_t start = lo ? fv->metadata.data[lo - 1].sequenceEndPosition : 0;
    out->sequenceIndex = lo;
    out->positionInSequence = globalPosition - start;
    return true;
}

void fastaVectorGetHeader(const FastaVector *fv, size_t index, const char **header, size_t *length) {
    size_t start = index ? fv->metadata.data[index - 1].headerEndPosition : 0;
    *header = fv->header.charData + start;
    *length = fv->metadata.data[index].headerEndPosition - start - 1;
}

void fastaVectorGetSequence(const FastaVector *fv, size_t index, const char **sequence, size_t *length) {
    size_t start = index ? fv->metadata.data[index - 1].sequenceEndPosition : 0;
    *sequence = fv->sequence.charData + start;
    *length = fv->metadata.data[index].sequenceEndPosition - start - 1;
}

}  // extern "C"
