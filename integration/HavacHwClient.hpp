// HavacHwClient.hpp -- the reference-side binding: a drop-in for the reference's host/HavacHwClient.hpp:22-75
// (and host/HavacHwClient.cpp:25-202, which it makes unnecessary) on top of the C ABI in include/havac_dev.h.
//
// A maintainer of the reference replaces host/HavacHwClient.{hpp,cpp} by this one file and links libhavac_dev.so
// instead of xrt_coreutil / uuid / xilinxopencl (CMakeLists.txt:61-66).  host/Havac.cpp, host/phmm/PhmmPreprocessor.cpp,
// host/sequence/SequencePreprocessor.cpp and PhmmReprojection/PhmmReprojection.cpp compile against it UNCHANGED:
// tests/refhost/Makefile does exactly that, from where those files lie, and tests/test_gpu_refhost.py runs the result
// on the GPU next to this repository's own Havac.
//
// Class name, method names, argument types, return types and exception types are the reference's.  `ert_cmd_state`
// is XRT's enum (the reference gets it from xrt/xrt_kernel.h); its values are the ones host/Havac.hpp:16-26 repeats.
// The reference's Havac.cpp uses std::cerr (host/Havac.cpp:178) and std::chrono without including <iostream> or
// <chrono> itself -- it receives both through the XRT headers of the file this one replaces, so they are included here.
#pragma once

#include <chrono>
#include <cstdint>
#include <iostream>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "havac_dev.h"

using std::shared_ptr;
using std::vector;

enum ert_cmd_state {   // numerically HAVAC_STATE_* of havac_dev.h
    ERT_CMD_STATE_NEW = 1,
    ERT_CMD_STATE_QUEUED = 2,
    ERT_CMD_STATE_RUNNING = 3,
    ERT_CMD_STATE_COMPLETED = 4,
    ERT_CMD_STATE_ERROR = 5,
    ERT_CMD_STATE_ABORT = 6,
    ERT_CMD_STATE_SUBMITTED = 7,
    ERT_CMD_STATE_TIMEOUT = 8,
    ERT_CMD_STATE_NORESPONSE = 9
};

class HavacHwClient {
public:
    // host/HavacHwClient.cpp:25-31.  No bitstream: the two strings are accepted and ignored.
    HavacHwClient(const std::string & /*xclbinFileSrc*/, const std::string & /*havacKernelName*/,
                  const uint32_t deviceIndex = 0) {
        const int rc = havac_dev_create(deviceIndex, &dev);
        if (rc == HAVAC_E_NOMEM) throw std::bad_alloc();
        if (rc != HAVAC_OK) throw std::runtime_error("HavacHwClient: no usable MI355X (gfx950) device at this index");
    }
    ~HavacHwClient() { havac_dev_destroy(dev); }
    HavacHwClient(HavacHwClient &&hc) = delete;
    HavacHwClient(HavacHwClient &hc) = delete;

    // host/HavacHwClient.cpp:78-110
    void writeSequence(const vector<uint8_t> &compressedSequence) {
        check(havac_dev_write_sequence(dev, compressedSequence.data(), compressedSequence.size()));
    }
    // host/HavacHwClient.cpp:111-138
    void writePhmm(shared_ptr<vector<int8_t>> phmmAsFlattenedArray) {
        check(havac_dev_write_phmm(dev, phmmAsFlattenedArray->data(), phmmAsFlattenedArray->size()));
    }
    // host/HavacHwClient.cpp:141-151
    void invokeHavacSsvAsync() { check(havac_dev_run_async(dev)); }
    // host/HavacHwClient.cpp:163-170
    ert_cmd_state getHwState() { return (ert_cmd_state)check(havac_dev_state(dev)); }
    // host/HavacHwClient.cpp:153-157
    ert_cmd_state waitForHavacSsvAsync(const std::chrono::milliseconds &timeout = std::chrono::milliseconds{0}) {
        return (ert_cmd_state)check(havac_dev_wait(dev, (uint32_t)timeout.count()));
    }
    // host/HavacHwClient.cpp:159-161
    ert_cmd_state abort() { return (ert_cmd_state)check(havac_dev_abort(dev)); }
    // host/HavacHwClient.cpp:172-202
    vector<uint64_t> getHitList() {
        uint32_t numHits = 0;
        check(havac_dev_num_hits(dev, &numHits));
        vector<uint64_t> hitsAsU64(numHits);
        if (numHits) check(havac_dev_read_hits(dev, hitsAsU64.data(), numHits));
        return hitsAsU64;
    }
    // ---- optional, no counterpart in the reference (one run at a time there): several runs open at once.  All the methods above
    // then speak of the OLDEST open run; retire() closes it.  Unused, nothing changes for the reference's Havac.cpp.
    void setPipelineDepth(uint32_t depth) { check(havac_dev_set_pipeline_depth(dev, depth)); }
    void retire() { check(havac_dev_retire(dev)); }
    uint32_t openRuns() { return havac_dev_open_runs(dev); }

private:
    // C code -> the exception type host/HavacHwClient.cpp throws at the corresponding place (include/havac_dev.h
    // lists the line per code); the message is the reference's, kept by the library.
    int check(int rc) {
        if (rc >= 0) return rc;
        const char *message = havac_dev_last_error(dev);
        if (rc == HAVAC_E_LENGTH) throw std::length_error(message);
        if (rc == HAVAC_E_LOGIC) throw std::logic_error(message);
        if (rc == HAVAC_E_NOMEM) throw std::bad_alloc();
        throw std::runtime_error(message);
    }
    havac_dev *dev = nullptr;
};
