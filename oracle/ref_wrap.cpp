/*
 * ref_wrap.cpp -- C entry point around the REFERENCE's softSsvThreshold256.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; the function it calls is the
 * reference's own test/softSsv/SoftSsv.cpp, compiled unmodified from where it
 * lies under /root/reference by oracle/Makefile (target `_ref`) into
 * oracle/_ref/libsoftssv_ref.so.  No reference source is copied into this
 * repository.  The library is used to validate oracle/ssv_oracle.c, to generate
 * tests/golden/ and (optionally) as bench.py's cpu_baseline of kind
 * "reference".
 */
#include <cstdint>
#include <vector>

#include "SoftSsv.h" /* from -I$(REFERENCE)/test/softSsv */

#ifndef HAVAC_BUILD_FLAGS
#define HAVAC_BUILD_FLAGS "?"
#endif

extern "C" {

/* compiler and flags this object -- the reference's SoftSsv.cpp with it -- was built with (bench.py: cpu_baseline.compiler;
 * SURVEY.md section 8d asks for it next to the core count) */
const char *softssv_ref_build_info(void) { return "g++ " __VERSION__ " " HAVAC_BUILD_FLAGS; }

/* Runs the reference on one-symbol-per-byte input and returns the number of
 * hits.  Up to `cap` hits are written as (row << 32 | column) in the
 * reference's own emission order (rows ascending, columns descending:
 * test/softSsv/SoftSsv.cpp:31-32).  Returns -1 when the reference reported
 * its allocation failure (errorCode -1, SoftSsv.cpp:24-29). */
int64_t softssv_ref_run(const uint8_t *symbols, uint64_t n, const int8_t *model, uint64_t nrows,
                        uint64_t *hits_row_col, uint64_t cap) {
    int8_t error = 0;
    std::vector<SoftSsvHit> found = softSsvThreshold256(symbols, n, model, (size_t)nrows, error);
    if (error != 0) return -1;
    uint64_t stored = 0;
    for (const SoftSsvHit &h : found) {
        if (stored >= cap) break;
        hits_row_col[stored++] = ((uint64_t)h.phmmPosition << 32) | (uint64_t)h.sequencePosition;
    }
    return (int64_t)found.size();
}

}
