/*
 * ssv_ref2_wrap.cpp -- C entry point around the REFERENCE's second CPU SSV, HitsFromSsv (host/test/Ssv.cpp:8-68): the
 * function the reference's on-FPGA test compares the device with (host/test/RefernceComparisonTest/
 * ReferenceComparisonTest.cpp:52-128) and the one boundary mode claims to match.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; HitsFromSsv, findThreshold256ScalingFactor and
 * emissionScoreToProjectedScore are the reference's, compiled from where they lie by tests/refhost/Makefile into
 * tests/_refhost/libssv_ref2.so, against the product's FastaVector.h / p7HmmReader.h (the reference's reader libraries
 * are un-vendored: a differential build, as for the projection).  Used in the build container only, by
 * tests/golden/make_golden_g8.py, to capture tests/golden/g8_boundary_*.npz.
 */
#include <cstdint>
#include <iostream>
#include <vector>

#include "Ssv.hpp" /* the reference's, -I$(REFERENCE)/host/test */

extern "C" {

/* Reads both files with the product's readers, runs HitsFromSsv, returns its hits in its own emission order
 * (model, record, position, row).  Call with cap = 0 to learn the count.  Returns 0, or -1 / -2 when the FASTA / the
 * .hmm file cannot be read. */
int ssv_ref2_hits(const char *fasta_path, const char *hmm_path, float p_value, uint32_t *sequence_number,
                  uint32_t *phmm_number, uint32_t *sequence_position, uint32_t *phmm_position, uint64_t cap,
                  uint64_t *count) {
    FastaVector fv;
    P7HmmList list;
    fastaVectorInit(&fv);
    if (fastaVectorReadFasta(fasta_path, &fv) != FASTA_VECTOR_OK) return -1;
    if (readP7Hmm(hmm_path, &list) != p7HmmSuccess) {
        fastaVectorDealloc(&fv);
        return -2;
    }
    std::cout.setstate(std::ios_base::failbit); /* HitsFromSsv prints a debug line per (model, record) */
    shared_ptr<vector<ReferenceSsvHit>> hits = HitsFromSsv(&fv, &list, p_value);
    std::cout.clear();
    *count = hits->size();
    for (uint64_t i = 0; i < cap && i < hits->size(); i++) {
        sequence_number[i] = (*hits)[i].sequenceNumber;
        phmm_number[i] = (*hits)[i].phmmNumber;
        sequence_position[i] = (*hits)[i].sequencePosition;
        phmm_position[i] = (*hits)[i].phmmPosition;
    }
    p7HmmListDealloc(&list);
    fastaVectorDealloc(&fv);
    return 0;
}

}
