// Havac.hpp -- the reference's public C++ API (host/Havac.hpp:16-107) on an MI355X.
//
// Drop-in: same enum, same classes, same method names, argument meaning and exception
// types.  What changed underneath: the XRT/HavacHwClient device path is gone; this
// class talks to libhavac_dev.so through the C ABI in include/havac_dev.h, and the
// un-vendored FastaVector / P7HmmReader libraries are replaced by the small readers
// in this directory.  No XRT, boost or Vitis header is needed to include this file.
#ifndef HAVAC_HOST_HPP
#define HAVAC_HOST_HPP

#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "FastaVector.h"
#include "p7HmmReader.h"

using std::shared_ptr;
using std::vector;

// host/Havac.hpp:16-26 (numerically XRT's ert_cmd_state)
enum havac_cmd_state {
    HAVAC_CMD_STATE_NEW = 1,
    HAVAC_CMD_STATE_QUEUED = 2,
    HAVAC_CMD_STATE_RUNNING = 3,
    HAVAC_CMD_STATE_COMPLETED = 4,
    HAVAC_CMD_STATE_ERROR = 5,
    HAVAC_CMD_STATE_ABORT = 6,
    HAVAC_CMD_STATE_SUBMITTED = 7,
    HAVAC_CMD_STATEIMEOUT = 8,      // (sic) the reference's spelling
    HAVAC_CMD_STATE_NORESPONSE = 9
};

// kept for source compatibility; there is no bitstream, the argument is ignored
const std::string xclbinSrcDefault = "../device/bin/havac.xclbin";

class HavacHit {   // host/Havac.hpp:31-39
public:
    HavacHit(const uint64_t sequencePosition, const uint32_t sequenceIndex, const uint32_t phmmPosition,
             const uint32_t phmmIndex);
    uint64_t sequencePosition;
    uint32_t sequenceIndex;
    uint32_t phmmPosition;
    uint32_t phmmIndex;
    // addition: set for hits found on the reverse-complemented record (Havac::setBothStrands); sequencePosition is
    // then still a position on the record as it stands in the file (the residue the hit cell read, complemented)
    bool reverseStrand = false;
    std::string toString();
};

// Addition (second half): what a long-target pipeline does with SSV hits before the
// next filter stage -- every hit becomes the stretch of its record the model would cover along the hit's
// diagonal, stretched by a flank, and overlapping stretches of the same (record, strand, model) are merged
// (nhmmer's window list after its SSV stage; the reference stops at the hit list and its validation tool only
// tests hits against nhmmer's envelopes, test/hmmerValidation/hmmerValidation.cpp:77-132).
struct HavacWindow {
    uint32_t sequenceIndex;
    uint32_t phmmIndex;
    bool reverseStrand;
    uint64_t sequenceStart, sequenceEnd;   // inclusive, positions on the record as it stands in the file
    uint32_t phmmFirst, phmmLast;          // lowest / highest model position among the window's hits
    uint32_t hitCount;
};

struct havac_dev;   // include/havac_dev.h

class Havac {       // host/Havac.hpp:42-107
public:
    Havac(const uint32_t deviceIndex = 0, const float requiredPValue = 0.02f,
          const std::string xclbinSrc = xclbinSrcDefault);
    // addition: one object over several GPUs of a node, one column shard each (the reference has one deviceIndex per
    // object); everything else behaves the same, getHitsFromFinishedRun returns the same list in the same order
    Havac(const std::vector<uint32_t> &deviceIndices, const float requiredPValue = 0.02f);
    // addition: the same object with the device layer starting on a helper thread.  Starting the HIP runtime takes
    // ~0.1 s whatever is asked of it; with this constructor that time runs under the caller's next calls -- loadPhmm and
    // loadSequence read and prepare their files on the host first and wait for the device only when they have something
    // to send.  A device that cannot be opened surfaces from the first call that needs it, as the exception the
    // reference-shaped constructor would have thrown (std::runtime_error / std::bad_alloc).
    struct DeferredStart {};
    Havac(DeferredStart, const uint32_t deviceIndex = 0, const float requiredPValue = 0.02f);
    Havac(Havac &&havac) = delete;
    Havac(Havac &havac) = delete;
    ~Havac();

    void loadSequence(const std::string fastaSrc);
    void loadPhmm(const std::string phmmSrc);
    void runHardwareClient();
    void runHardwareClientAsync();
    void waitHardwareClientAsync();
    void abortHardwareClient();
    vector<HavacHit> getHitsFromFinishedRun();
    enum havac_cmd_state currentHardwareState();

    // ---- additions (not in the reference) ----
    // Boundary mode: every (model, record) pair is scored on its own, as the
    // reference's second CPU SSV does (host/test/Ssv.cpp:8-68), instead of one diagonal sweep over the
    // concatenation.  Call before loadPhmm / loadSequence.  Default off = the reference's device semantics.
    void setBoundaryMode(bool on);
    // Both strands: also scores the reverse complement of every record, which nhmmer
    // does by default and the reference's benchmark switched off (--watson, benchmark/readme.txt:62-64).  Hits of
    // that half carry reverseStrand = true.  Call before loadSequence.  Default off.
    void setBothStrands(bool on);
    // Hits of the finished run merged into windows (see HavacWindow); `flank` residues are added on both sides.
    vector<HavacWindow> getWindowsFromFinishedRun(uint32_t flank = 0);
    // Packing on the GPU, on by default: loadSequence sends the text (and, in the plain
    // mode, the symbols it drew for the non-a/c/g/t columns); the boundary-mode layout and the second strand are made
    // on the GPU as well.  The device buffers are byte for byte what the host packer (SequencePreprocessor) would have
    // produced with the same rand() state.  Off = pack on the host.
    void setDevicePacking(bool on);
    // Runs in flight (the reference's runHardwareClient is one run at a time, host/Havac.cpp:80-98): with a depth above 1
    // runHardwareClientAsync may be called again -- after loadPhmm of the next model file, say -- before the hits of the run
    // before are fetched.  The records of run k are then ordered and read back while the kernel of run k + 1 runs
    // (include/havac_dev.h: havac_dev_set_pipeline_depth).  wait / state / abort / getHitsFromFinishedRun speak of the OLDEST run
    // whose hits have not been fetched; getHitsFromFinishedRun waits for it if need be, resolves its hits against the models IT
    // ran with, and closes it.  Depth 1 (the default) is the reference's behaviour in every respect.  Call between runs.
    void setPipelineDepth(uint32_t depth);
    void setHitCapacity(uint64_t maxHits);                 // the reference's buffer is a fixed 3.5 GiB
    void lastRunMilliseconds(float *ssvKernelMs, float *totalMs);
    const vector<uint64_t> &rawHitsOfLastFetch() const { return rawHits_; }

private:
    void check(int code);                                  // C-ABI code -> the reference's exception types
    void init();
    void needDevice();                                     // joins a deferred start; throws what the constructor would have
    std::thread deviceStart_;
    int deviceStartCode_ = 0;
    vector<uint32_t> generatePhmmLenPrefixSums();

    havac_dev *dev_ = nullptr;
    FastaVector *fastaVector = nullptr;
    P7HmmList *p7HmmList = nullptr;
    shared_ptr<vector<int8_t>> compressedPhmmScores;
    uint32_t deviceIndex;
    float requiredPValue;
    bool phmmLoadedToDevice = false;
    bool sequenceLoadedToDevice = false;
    vector<uint64_t> rawHits_;
    bool boundaryMode_ = false;
    bool bothStrands_ = false;
    bool devicePacking_ = true;
    uint64_t forwardColumns_ = 0;                          // both strands: columns of the forward half
    vector<uint64_t> residueCounts_;                       // both strands: residues per record
    vector<uint64_t> recordStarts_, recordLengths_;        // boundary mode: global column of each record
    vector<uint32_t> modelStarts_;                         // boundary mode: global row of each model
    // what a run's hits are resolved against: the models it was started with (loadPhmm may have replaced them since)
    struct RunModels { vector<uint32_t> prefixSums, lengths, starts; };
    std::deque<RunModels> runModels_;                      // one per open run, oldest first
    uint32_t pipelineDepth_ = 1;
    vector<HavacHit> fetchHits(vector<uint32_t> *modelLengthsOut);
};

// The resolver of host/Havac.cpp:145-187 as a free function (testable without a device):
// raw 64-bit records -> HavacHit, dropping hits in the padding after the last record.
struct PhmmLocalPosition { int32_t phmmIndex; uint32_t phmmPosition; };
PhmmLocalPosition phmmPrefixSumsBinarySearch(uint32_t phmmGlobalPosition, vector<uint32_t> &prefixSums);
vector<HavacHit> havacResolveHits(const vector<uint64_t> &rawHits, const FastaVector *fastaVector,
                                  vector<uint32_t> &phmmPrefixSums);

// Hits -> merged windows, as a free function (testable without a device).  A hit at model position k (0-based, model
// length L) and record position i covers [i - k, i + (L-1-k)] on the forward strand and [i - (L-1-k), i + k] on the
// reverse strand (there the model runs towards lower file positions); the stretch is widened by `flank`, clipped to
// the record [0, recordLength), and stretches of one (record, strand, model) that overlap or touch become one window.
// Output order: record, strand (forward first), model, start.
vector<HavacWindow> havacMergeHitsToWindows(const vector<HavacHit> &hits, const vector<uint32_t> &modelLengths,
                                            const vector<uint64_t> &recordLengths, uint32_t flank = 0);
#endif
