// FastaVector.h -- minimal FASTA store with the surface HAVAC's host code uses.
//
// The reference depends on the un-vendored TravisWheelerLab/FastaVector C library
// (.gitmodules:1-3; the directory is empty in the reference tree), through
// exactly these members:
//   sequence.charData / sequence.count   host/sequence/SequencePreprocessor.cpp:10-13,43-44
//   metadata.count / data[i].sequenceEndPosition
//   fastaVectorInit / ReadFasta / Dealloc          host/Havac.cpp:27-30,34,58-67
//   fastaVectorGetLocalSequencePositionFromGlobal  host/Havac.cpp:165-167
// This is our own reader with the same names and meaning.  Definitions we had
// to choose because the library is absent (pinned by our own tests only):
//   * every record's residues are followed by one '\0' in charData, and
//     sequence.count includes those terminators (the reference relies on this:
//     "the sequences are seperated by null terminators",
//     host/sequence/SequencePreprocessor.cpp:12);
//   * sequenceEndPosition(i) is one past that terminator, so records tile
//     [0, sequence.count) and a global position on a terminator resolves to
//     its record with positionInSequence == the record's length.
#ifndef HAVAC_FASTA_VECTOR_H
#define HAVAC_FASTA_VECTOR_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum FastaVectorReturnCode {
    FASTA_VECTOR_OK = 0,
    FASTA_VECTOR_FILE_OPEN_FAIL = 1,
    FASTA_VECTOR_FILE_READ_FAIL = 2,
    FASTA_VECTOR_ALLOCATION_FAIL = 3
};

struct FastaVectorString { char *charData; size_t count; size_t capacity; };
struct FastaVectorMetadata { size_t headerEndPosition; size_t sequenceEndPosition; };
struct FastaVectorMetadataVector { struct FastaVectorMetadata *data; size_t count; size_t capacity; };

struct FastaVector {
    struct FastaVectorString sequence;   /* residues of all records, '\0' after each */
    struct FastaVectorString header;     /* header lines (without '>'), '\0' after each */
    struct FastaVectorMetadataVector metadata;
};

struct FastaVectorLocalPosition { size_t sequenceIndex; size_t positionInSequence; };

enum FastaVectorReturnCode fastaVectorInit(struct FastaVector *fv);
void fastaVectorDealloc(struct FastaVector *fv);
/* appends the records of the file to whatever the vector already holds */
enum FastaVectorReturnCode fastaVectorReadFasta(const char *path, struct FastaVector *fv);
enum FastaVectorReturnCode fastaVectorAddSequenceToList(struct FastaVector *fv, const char *header, size_t headerLength,
                                                        const char *sequence, size_t sequenceLength);
bool fastaVectorGetLocalSequencePositionFromGlobal(const struct FastaVector *fv, size_t globalPosition,
                                                   struct FastaVectorLocalPosition *out);
void fastaVectorGetHeader(const struct FastaVector *fv, size_t index, const char **header, size_t *length);
void fastaVectorGetSequence(const struct FastaVector *fv, size_t index, const char **sequence, size_t *length);

#ifdef __cplusplus
}
#endif
#endif
