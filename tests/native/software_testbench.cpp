// software_testbench.cpp -- the reference's HLS C-simulation testbench (device/test/softwareTestbench.cpp:49-306),
// with the MI355X device layer in the place of the C-simulated HavacKernel and the CPU checker in the place of the
// reference's softSsv object (which cannot travel to the GPU box; oracle/ssv_oracle.c is proven equal to it in the
// build container).  TEST CODE: links liboracle.so.
//
// Same flow as the reference's main(): per test generate a sequence and a model, unpack the sequence to one symbol
// per byte (:324-336), run the software SSV (:109-110), pack 2-bit (:340-352), run the device (:200-201), decode the
// 64-bit hit reports (device/HavacHls.hpp:22-36), and compare the two hit lists as sets in both directions
// (:224-306).  Differences from the reference: fixed seeds instead of srand(time(0)) (:50); synthetic Dfam-like models
// instead of `hmmbuild` (test/generator/hmmSeqGenerator.cpp:128-132, not available); a non-zero exit code on
// mismatch (the reference's second comparison loop computes foundMatchingSoftwareHit and never fails on it, :265-303).
//
//   usage: software_testbench [numTests] [segments] [modelRows]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/havac_dev.h"
#include "../../oracle/ssv_oracle.h"

namespace {
const uint32_t kSegment = HAVAC_SEGMENT_COLUMNS;      // NUM_CELL_PROCESSORS

struct TestbenchHitReport {                           // device/HavacHls.hpp:22-36
    explicit TestbenchHitReport(uint64_t reportAsU64) {
        const uint64_t partition = reportAsU64 & ((1ULL << 14) - 1);
        const uint64_t segment = (reportAsU64 & ((1ULL << 40) - 1)) >> 14;
        sequencePosition = segment * (12 * 1024) + partition;
        phmmPosition = (uint32_t)(reportAsU64 >> 40);
    }
    uint64_t sequencePosition;
    uint32_t phmmPosition;
};

struct Hit { uint32_t phmmPosition; uint64_t sequencePosition; };
bool operator<(const Hit &a, const Hit &b) {
    return a.phmmPosition != b.phmmPosition ? a.phmmPosition < b.phmmPosition : a.sequencePosition < b.sequencePosition;
}
bool operator==(const Hit &a, const Hit &b) { return a.phmmPosition == b.phmmPosition && a.sequencePosition == b.sequencePosition; }

// every hit of `from` must be in `in`; prints the strays like the reference's compareSsvHitLists
bool allFound(const std::vector<Hit> &from, const std::vector<Hit> &in, const char *fromName, const char *inName) {
    bool ok = true;
    int shown = 0;
    for (const Hit &h : from) {
        if (!std::binary_search(in.begin(), in.end(), h)) {
            ok = false;
            if (shown++ < 10)
                std::printf("  %s hit (phmm %u, seq %llu) has no matching %s hit\n", fromName, h.phmmPosition,
                            (unsigned long long)h.sequencePosition, inName);
        }
    }
    return ok;
}
}  // namespace

int main(int argc, char **argv) {
    const uint32_t numTests = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2;                 // :44
    const uint32_t segments = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 3;                  // TEST_NUM_SEQUENCE_SEGMENTS
    const uint32_t modelRows = argc > 3 ? (uint32_t)std::atoi(argv[3]) : segments * kSegment;   // :65 model length = sequence length
    std::printf("BEGINNING TESTBENCH...\n");
    havac_dev *dev = nullptr;
    if (havac_dev_create(0, &dev) != HAVAC_OK) { std::printf("ERROR: no MI355X device\n"); return 3; }
    bool allTestsPassed = true;

    for (uint32_t test = 1; test <= numTests; test++) {
        std::printf(" Test #%u/%u \n", test, numTests);
        std::mt19937 rng(1000 + test);
        const uint64_t n = (uint64_t)segments * kSegment;
        // model: a consensus base scoring +20..+40, the others -60..-30; sequence: the consensus mutated at 30 %
        // (the reference mutates the generating sequence at 0.3, :54,:382-397) laid over a random background
        std::vector<int8_t> model((size_t)modelRows * 4);
        std::vector<uint8_t> consensus(modelRows);
        for (uint32_t r = 0; r < modelRows; r++) {
            consensus[r] = (uint8_t)(rng() & 3);
            for (int a = 0; a < 4; a++) model[(size_t)r * 4 + a] = (int8_t)(-60 + (int)(rng() % 31));
            model[(size_t)r * 4 + consensus[r]] = (int8_t)(20 + (int)(rng() % 21));
        }
        std::vector<uint8_t> sequenceAsVectorIndices(n);
        for (uint64_t s = 0; s < n; s++) {
            const bool mutate = (rng() % 10) < 3;
            sequenceAsVectorIndices[s] = (s < modelRows && !mutate) ? consensus[s] : (uint8_t)(rng() & 3);
        }

        // software SSV on one symbol per byte
        std::vector<uint64_t> soft(1 << 20);
        int64_t nsoft = havac_oracle_ssv(sequenceAsVectorIndices.data(), n, model.data(), modelRows, soft.data(), soft.size());
        if (nsoft < 0 || (uint64_t)nsoft > soft.size()) { std::printf("Error: soft ssv returned %lld\n", (long long)nsoft); return 8; }
        std::printf("softSsv ran\n");

        // pack the sequence into 2-bit values for the device
        std::vector<uint8_t> packed(n / 4);
        havac_oracle_pack_2bit(sequenceAsVectorIndices.data(), n, packed.data());

        std::printf("invoking hardware ssv...\n");
        int rc = havac_dev_write_sequence(dev, packed.data(), packed.size());
        if (rc == HAVAC_OK) rc = havac_dev_write_phmm(dev, model.data(), model.size());
        if (rc == HAVAC_OK) rc = havac_dev_run_async(dev);
        if (rc != HAVAC_OK || havac_dev_wait(dev, 0) != HAVAC_STATE_COMPLETED) {
            std::printf("ERROR: device run failed: %s\n", havac_dev_last_error(dev));
            return 9;
        }
        uint32_t nhw = 0;
        havac_dev_num_hits(dev, &nhw);
        std::vector<uint64_t> raw(nhw);
        if (nhw) havac_dev_read_hits(dev, raw.data(), nhw);

        std::vector<Hit> softwareSsvHits, hardwareSsvHits;
        for (int64_t i = 0; i < nsoft; i++) {
            TestbenchHitReport r(soft[(size_t)i]);
            softwareSsvHits.push_back({r.phmmPosition, r.sequencePosition});
        }
        for (uint64_t rep : raw) {
            TestbenchHitReport r(rep);
            hardwareSsvHits.push_back({r.phmmPosition, r.sequencePosition});
        }
        std::sort(softwareSsvHits.begin(), softwareSsvHits.end());
        std::sort(hardwareSsvHits.begin(), hardwareSsvHits.end());
        std::printf("comparing %zu software hits with %zu hardware hits\n", softwareSsvHits.size(), hardwareSsvHits.size());
        const bool a = allFound(hardwareSsvHits, softwareSsvHits, "hardware", "software");
        const bool b = allFound(softwareSsvHits, hardwareSsvHits, "software", "hardware");
        const bool passed = a && b && softwareSsvHits.size() == hardwareSsvHits.size();
        std::printf("test %u %s\n", test, passed ? "passed" : "FAILED");
        allTestsPassed = allTestsPassed && passed;
    }
    havac_dev_destroy(dev);
    std::printf(allTestsPassed ? "ALL TESTS PASSED\n" : "SOME TESTS FAILED\n");
    return allTestsPassed ? 0 : 1;
}
