// p7HmmReader.h -- minimal HMMER3 ASCII profile reader with the surface HAVAC uses.
//
// The reference depends on the un-vendored TravisWheelerLab/P7HmmReader C
// library (.gitmodules:4-6; empty in the reference tree) through exactly these
// members:
//   readP7Hmm(path, &list) and its return codes        host/Havac.cpp:43-50
//   P7HmmList{phmms, count}; p7HmmListDealloc          host/Havac.cpp:35,111
//   header.modelLength / maxLength                     PhmmReprojection.cpp:39-40,110
//   stats.msvGumbelMu / msvGumbelLambda                PhmmReprojection.cpp:37-38
//   model.matchEmissionScores (float[L*K], file values = -ln p)   PhmmReprojection.cpp:133
//   p7HmmGetAlphabetCardinality                        PhmmReprojection.cpp:117
// This is our own parser with the same names.  Nothing in the reference pins
// parser behaviour; ours is pinned by tests/test_host_readers.py.
#ifndef HAVAC_P7_HMM_READER_H
#define HAVAC_P7_HMM_READER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum P7HmmReturnCode {
    p7HmmSuccess = 0,
    p7HmmFileNotFound = 1,
    p7HmmAllocationFailure = 2,
    p7HmmFormatError = 3
};

enum P7Alphabet { P7HmmReaderAlphabetAmino = 0, P7HmmReaderAlphabetDna = 1, P7HmmReaderAlphabetRna = 2 };

struct P7Header {
    char *name;              /* NAME */
    char *accessionNumber;   /* ACC, may be NULL */
    uint32_t modelLength;    /* LENG */
    uint32_t maxLength;      /* MAXL */
    enum P7Alphabet alphabet;/* ALPH */
};

struct P7Stats {
    float msvGumbelMu, msvGumbelLambda;        /* STATS LOCAL MSV */
    float viterbiGumbelMu, viterbiGumbelLambda;/* STATS LOCAL VITERBI */
    float forwardTau, forwardLambda;           /* STATS LOCAL FORWARD */
    int hasMsv;
};

struct P7Model {
    float *matchEmissionScores;   /* [modelLength][K], -ln(probability); '*' -> +infinity */
};

struct P7Hmm {
    struct P7Header header;
    struct P7Stats stats;
    struct P7Model model;
};

struct P7HmmList {
    struct P7Hmm *phmms;
    uint32_t count;
};

enum P7HmmReturnCode readP7Hmm(const char *path, struct P7HmmList *list);
void p7HmmListDealloc(struct P7HmmList *list);
uint32_t p7HmmGetAlphabetCardinality(const struct P7Hmm *phmm);

#ifdef __cplusplus
}
#endif
#endif
