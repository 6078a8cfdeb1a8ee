// Havac.cpp -- host/Havac.cpp:20-206 on top of the C ABI of libhavac_dev.so.
#include "Havac.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <new>
#include <sstream>
#include <stdexcept>

#include "../../../include/havac_dev.h"
#include "HostThreads.hpp"
#include "PhmmPreprocessor.hpp"
#include "SequencePreprocessor.hpp"

// ---- error mapping (include/havac_dev.h names the reference's exception per code) ----
void Havac::check(int code) {
    if (code >= 0) return;
    const char *msg = dev_ ? havac_dev_last_error(dev_) : "";
    switch (code) {
        case HAVAC_E_LENGTH: throw std::length_error(msg);
        case HAVAC_E_LOGIC: throw std::logic_error(msg);
        case HAVAC_E_NOMEM: throw std::bad_alloc();
        case HAVAC_E_HIT_OVERFLOW: throw std::overflow_error(msg);
        default: throw std::runtime_error(msg && *msg ? msg : "havac device error");
    }
}

Havac::Havac(const uint32_t deviceIndex, const float requiredPValue, const std::string)
    : deviceIndex(deviceIndex), requiredPValue(requiredPValue) {
    int rc = havac_dev_create(deviceIndex, &dev_);
    if (rc == HAVAC_E_NOMEM) throw std::bad_alloc();
    if (rc != HAVAC_OK) throw std::runtime_error("ERROR: could not open MI355X device " + std::to_string(deviceIndex));
    init();
}

Havac::Havac(const std::vector<uint32_t> &deviceIndices, const float requiredPValue)
    : deviceIndex(deviceIndices.empty() ? 0 : deviceIndices[0]), requiredPValue(requiredPValue) {
    int rc = havac_dev_create_multi(deviceIndices.data(), (uint32_t)deviceIndices.size(), &dev_);
    if (rc == HAVAC_E_NOMEM) throw std::bad_alloc();
    if (rc != HAVAC_OK) throw std::runtime_error("ERROR: could not open the requested MI355X devices");
    init();
}

Havac::Havac(DeferredStart, const uint32_t deviceIndex, const float requiredPValue)
    : deviceIndex(deviceIndex), requiredPValue(requiredPValue) {
    init();
    deviceStart_ = std::thread([this, deviceIndex] { deviceStartCode_ = havac_dev_create(deviceIndex, &dev_); });
}

void Havac::needDevice() {
    if (deviceStart_.joinable()) deviceStart_.join();
    if (deviceStartCode_ == HAVAC_E_NOMEM) throw std::bad_alloc();
    if (deviceStartCode_ != HAVAC_OK) throw std::runtime_error("ERROR: could not open MI355X device " + std::to_string(deviceIndex));
}

void Havac::init() {
    fastaVector = static_cast<FastaVector *>(std::malloc(sizeof(FastaVector)));
    p7HmmList = static_cast<P7HmmList *>(std::calloc(1, sizeof(P7HmmList)));
    if (!fastaVector || !p7HmmList || fastaVectorInit(fastaVector) == FASTA_VECTOR_ALLOCATION_FAIL) {
        std::free(fastaVector); std::free(p7HmmList);
        havac_dev_destroy(dev_);
        throw std::bad_alloc();
    }
}

Havac::~Havac() {
    if (deviceStart_.joinable()) deviceStart_.join();
    fastaVectorDealloc(fastaVector);
    p7HmmListDealloc(p7HmmList);
    std::free(fastaVector);
    std::free(p7HmmList);
    havac_dev_destroy(dev_);
}

void Havac::loadPhmm(const std::string phmmSrc) {
    p7HmmListDealloc(p7HmmList);
    enum P7HmmReturnCode rc = readP7Hmm(phmmSrc.c_str(), p7HmmList);
    // host/Havac.cpp:44-50.  The reference's test for p7HmmFileNotFound is a dangling `if` that
    // falls through to preprocessing an uninitialised list; here a missing file is an error.
    if (rc == p7HmmFileNotFound) throw std::runtime_error("Could not open phmm file for reading.");
    if (rc == p7HmmAllocationFailure) throw std::bad_alloc();
    if (rc == p7HmmFormatError) throw std::runtime_error("Phmm file was not formatted correctly.");
    PhmmPreprocessor preprocessor(p7HmmList, requiredPValue, boundaryMode_);
    compressedPhmmScores = preprocessor.getProcessedPhmmData();
    modelStarts_ = preprocessor.getModelStarts();
    needDevice();
    check(havac_dev_write_phmm(dev_, compressedPhmmScores->data(), compressedPhmmScores->size()));
    phmmLoadedToDevice = true;
}

void Havac::loadSequence(const std::string fastaSrc) {
    enum FastaVectorReturnCode rc = fastaVectorReadFasta(fastaSrc.c_str(), fastaVector);   // appends, as the reference does
    if (rc == FASTA_VECTOR_ALLOCATION_FAIL) throw std::bad_alloc();
    if (rc == FASTA_VECTOR_FILE_OPEN_FAIL) throw std::runtime_error("Could not open fasta file for reading.");
    if (rc == FASTA_VECTOR_FILE_READ_FAIL) throw std::runtime_error("Error while reading from the opened fasta file.");
    if (devicePacking_) {
        // the text goes to the GPU and every layout is made there.
        vector<uint64_t> ends, starts, residues;
        for (size_t j = 0; j < fastaVector->metadata.count; j++) ends.push_back(fastaVector->metadata.data[j].sequenceEndPosition);
        if (boundaryMode_) {
            needDevice();
            // every record its own columns + a separator pair, a/c/g as they are, everything else T (no rand())
            recordStarts_.assign(ends.size(), 0);
            check(havac_dev_write_sequence_records(dev_, fastaVector->sequence.charData, fastaVector->sequence.count, ends.data(),
                                                   (uint32_t)ends.size(), recordStarts_.data()));
            recordLengths_.clear();
            for (size_t j = 0; j < ends.size(); j++) recordLengths_.push_back(ends[j] - (j ? ends[j - 1] : 0));
        } else {
            // the reference's layout: the host only looks for the characters that are not a/c/g/t and draws their
            // symbols (same rand() order as SequencePreprocessor); chunks are packed while later ones cross PCIe
            vector<uint64_t> patchColumns;
            vector<uint8_t> patchSymbols;
            SequencePreprocessor::collectPatches(fastaVector, patchColumns, patchSymbols);
            needDevice();
            check(havac_dev_write_sequence_chars(dev_, fastaVector->sequence.charData, fastaVector->sequence.count,
                                                 patchColumns.data(), patchSymbols.data(), patchColumns.size()));
        }
        if (bothStrands_) {
            for (size_t j = 0; j < ends.size(); j++) {
                const uint64_t begin = j ? ends[j - 1] : 0;
                starts.push_back(boundaryMode_ ? recordStarts_[j] : begin);
                residues.push_back(ends[j] - begin - 1);
            }
            check(havac_dev_append_reverse_strand(dev_, starts.data(), residues.data(), (uint32_t)starts.size(), &forwardColumns_));
            residueCounts_ = residues;
        }
        sequenceLoadedToDevice = true;
        return;
    }
    SequencePreprocessor preprocessor(fastaVector, boundaryMode_);
    if (bothStrands_) {
        vector<uint64_t> starts, residues;
        for (size_t j = 0; j < fastaVector->metadata.count; j++) {
            const uint64_t begin = j ? fastaVector->metadata.data[j - 1].sequenceEndPosition : 0;
            starts.push_back(boundaryMode_ ? preprocessor.getRecordStarts()[j] : begin);
            residues.push_back(fastaVector->metadata.data[j].sequenceEndPosition - begin - 1);
        }
        forwardColumns_ = preprocessor.appendReverseStrand(starts, residues);
        residueCounts_ = residues;
    }
    vector<uint8_t> &packed = preprocessor.getCompressedSequenceBuffer();
    needDevice();
    check(havac_dev_write_sequence(dev_, packed.data(), packed.size()));
    if (boundaryMode_) {
        vector<uint8_t> &mask = preprocessor.getSeparatorMask();
        check(havac_dev_write_separator_mask(dev_, mask.data(), mask.size()));
        recordStarts_ = preprocessor.getRecordStarts();
        recordLengths_ = preprocessor.getRecordLengths();
    }
    sequenceLoadedToDevice = true;
}

void Havac::runHardwareClient() {
    runHardwareClientAsync();
    waitHardwareClientAsync();
}

void Havac::runHardwareClientAsync() {
    if (!phmmLoadedToDevice)   // host/Havac.cpp:86-88
        throw std::logic_error("Phmm was not loaded to device before hardware was requested to run.");
    if (!sequenceLoadedToDevice)   // :89-91
        throw std::logic_error("Sequence was not loaded to device before hardware was requested to run.");
    needDevice();
    check(havac_dev_run_async(dev_));
    // the models this run's hits belong to (host/Havac.cpp:104-116 makes the prefix sums when the hits are fetched; here the
    // model list may have been replaced by then)
    RunModels m;
    m.prefixSums = generatePhmmLenPrefixSums();
    for (uint32_t i = 0; i < p7HmmList->count; i++) m.lengths.push_back(p7HmmList->phmms[i].header.modelLength);
    m.starts = modelStarts_;
    if (pipelineDepth_ == 1) runModels_.clear();           // (the finished run before this one was closed by the device layer too)
    runModels_.push_back(std::move(m));
}

void Havac::setPipelineDepth(uint32_t depth) {
    needDevice();
    check(havac_dev_set_pipeline_depth(dev_, depth));
    pipelineDepth_ = depth;
    runModels_.clear();
}

void Havac::waitHardwareClientAsync() { needDevice(); check(havac_dev_wait(dev_, 0)); }

void Havac::abortHardwareClient() { needDevice(); check(havac_dev_abort(dev_)); }

enum havac_cmd_state Havac::currentHardwareState() {
    needDevice();
    int s = havac_dev_state(dev_);
    check(s);
    return (havac_cmd_state)s;
}

void Havac::setBoundaryMode(bool on) {
    if (phmmLoadedToDevice || sequenceLoadedToDevice)
        throw std::logic_error("setBoundaryMode must be called before loadPhmm and loadSequence.");
    boundaryMode_ = on;
}

void Havac::setDevicePacking(bool on) {
    if (sequenceLoadedToDevice) throw std::logic_error("setDevicePacking must be called before loadSequence.");
    devicePacking_ = on;
}

void Havac::setBothStrands(bool on) {
    if (sequenceLoadedToDevice) throw std::logic_error("setBothStrands must be called before loadSequence.");
    bothStrands_ = on;
}

void Havac::setHitCapacity(uint64_t maxHits) { needDevice(); check(havac_dev_set_hit_capacity(dev_, maxHits)); }

void Havac::lastRunMilliseconds(float *ssvKernelMs, float *totalMs) {
    needDevice();
    check(havac_dev_last_run_ms(dev_, ssvKernelMs, totalMs));
}

vector<uint32_t> Havac::generatePhmmLenPrefixSums() {   // host/Havac.cpp:104-116
    vector<uint32_t> sums(1, 0u);
    sums.reserve(p7HmmList->count + 1);
    for (uint32_t i = 0; i < p7HmmList->count; i++) sums.push_back(sums.back() + p7HmmList->phmms[i].header.modelLength);
    return sums;
}

// largest index whose prefix sum is <= the global row (host/Havac.cpp:119-142)
PhmmLocalPosition phmmPrefixSumsBinarySearch(uint32_t phmmGlobalPosition, vector<uint32_t> &prefixSums) {
    int32_t lo = 0, hi = (int32_t)prefixSums.size() - 1, found = -1;
    while (lo <= hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (prefixSums[mid] <= phmmGlobalPosition) { found = mid; lo = mid + 1; } else { hi = mid - 1; }
    }
    PhmmLocalPosition p;
    p.phmmIndex = found;
    p.phmmPosition = found >= 0 ? phmmGlobalPosition - prefixSums[found] : 0;
    return p;
}

// One raw record -> a HavacHit (host/Havac.cpp:150-184); false: a hit in the padding after the last record, dropped.
static bool resolveOne(uint64_t rec, size_t index, const FastaVector *fastaVector, vector<uint32_t> &phmmPrefixSums, HavacHit *out) {
    // [13:0] column in segment, [39:14] segment, [63:40] row (host/Havac.cpp:155-163)
    const uint64_t inSegment = rec & ((1ull << 14) - 1);
    const uint64_t segment = (rec & ((1ull << 40) - 1)) >> 14;
    const uint64_t globalSequencePosition = segment * (12 * 1024) + inSegment;
    const uint32_t globalPhmmPosition = (uint32_t)(rec >> 40);
    FastaVectorLocalPosition local;
    if (!fastaVectorGetLocalSequencePositionFromGlobal(fastaVector, globalSequencePosition, &local))
        return false;   // a hit in the padding after the last record (host/Havac.cpp:169-173)
    PhmmLocalPosition where = phmmPrefixSumsBinarySearch(globalPhmmPosition, phmmPrefixSums);
    if (where.phmmIndex == -1) {
        std::cerr << "ERROR: could not resolve phmm position for raw hit report #" << index << "\n" << std::endl;
        return false;
    }
    *out = HavacHit(local.positionInSequence, (uint32_t)local.sequenceIndex, where.phmmPosition, (uint32_t)where.phmmIndex);
    return true;
}

// fn(i, &hit) -> keep? for every i in [0, n), results in index order.  Lists of a run hold 10^5 ... 10^9 records and every
// record resolves on its own: long lists are cut into stretches resolved side by side on the host's cores.
template <class F>
static vector<HavacHit> resolveAll(size_t n, F &&fn) {
    const size_t kStretch = 1 << 15;
    const size_t stretches = (n + kStretch - 1) / kStretch;
    vector<vector<HavacHit>> parts(stretches);
    havacParallelFor(stretches, n < 4 * kStretch ? 1u : havacHostThreads(stretches), [&](size_t s) {
        const size_t begin = s * kStretch, end = std::min(n, begin + kStretch);
        vector<HavacHit> &mine = parts[s];
        mine.reserve(end - begin);
        HavacHit hit(0, 0, 0, 0);
        for (size_t i = begin; i < end; i++)
            if (fn(i, &hit)) mine.push_back(hit);
    });
    if (stretches == 1) return std::move(parts[0]);
    size_t total = 0;
    for (const auto &p : parts) total += p.size();
    vector<HavacHit> out;
    out.reserve(total);
    for (const auto &p : parts) out.insert(out.end(), p.begin(), p.end());
    return out;
}

vector<HavacHit> havacResolveHits(const vector<uint64_t> &rawHits, const FastaVector *fastaVector,
                                  vector<uint32_t> &phmmPrefixSums) {
    return resolveAll(rawHits.size(), [&](size_t i, HavacHit *hit) { return resolveOne(rawHits[i], i, fastaVector, phmmPrefixSums, hit); });
}

vector<HavacHit> Havac::getHitsFromFinishedRun() { return fetchHits(nullptr); }

vector<HavacHit> Havac::fetchHits(vector<uint32_t> *modelLengthsOut) {
    needDevice();
    uint64_t n = 0;                                        // 64-bit: several GPUs can hold more than 2^32 - 1 records
    check(havac_dev_num_hits64(dev_, &n));
    rawHits_.assign(n, 0);
    if (n) check(havac_dev_read_hits64(dev_, rawHits_.data(), n));
    // both strands: a record of the second half is the record at (column - forwardColumns_) of the first.  Folded record by
    // record where it is resolved: no second copy of the list and no flag per record (with one strand -- the reference's
    // mode -- nothing at all is done: C4's list is 36 GB)
    const bool bothStrands = bothStrands_;
    const uint64_t forwardColumns = forwardColumns_;
    auto fold = [bothStrands, forwardColumns](uint64_t rec, bool *isReverse) -> uint64_t {
        *isReverse = false;
        if (!bothStrands) return rec;
        uint64_t column = ((rec >> 14) & 0x3ffffffull) * 12288ull + (rec & 0x3fffull);
        if (column < forwardColumns) return rec;
        column -= forwardColumns;
        *isReverse = true;
        return (rec & ~((1ull << 40) - 1)) | ((column / 12288ull) << 14) | (column % 12288ull);
    };
    const vector<uint64_t> &raw = rawHits_;
    auto mirror = [&](HavacHit &h, bool isReverse) {
        if (!isReverse) return;
        h.reverseStrand = true;
        const uint64_t n = residueCounts_[h.sequenceIndex];
        if (h.sequencePosition < n) h.sequencePosition = n - 1 - h.sequencePosition;   // the terminator column stays
    };
    // the models this run ran with (a run started before loadPhmm replaced them; with nothing recorded: the current ones)
    RunModels models;
    if (!runModels_.empty()) models = runModels_.front();
    else {
        models.prefixSums = generatePhmmLenPrefixSums();
        for (uint32_t i = 0; i < p7HmmList->count; i++) models.lengths.push_back(p7HmmList->phmms[i].header.modelLength);
        models.starts = modelStarts_;
    }
    if (modelLengthsOut) *modelLengthsOut = models.lengths;
    // with several runs open, fetching a run's hits closes it: the next call speaks of the next run
    struct CloseRun {
        Havac *h;
        ~CloseRun() { if (h->pipelineDepth_ > 1) { (void)havac_dev_retire(h->dev_); if (!h->runModels_.empty()) h->runModels_.pop_front(); } }
    } closeRun{this};
    if (!boundaryMode_) {
        vector<uint32_t> &sums = models.prefixSums;
        return resolveAll(raw.size(), [&](size_t i, HavacHit *hit) {
            bool isReverse;
            if (!resolveOne(fold(raw[i], &isReverse), i, fastaVector, sums, hit)) return false;
            mirror(*hit, isReverse);
            return true;
        });
    }
    // boundary mode: records and models have their own start tables (separators in between)
    const vector<uint32_t> &modelStarts_ = models.starts;
    return resolveAll(raw.size(), [&](size_t i, HavacHit *hit) {
        bool isReverse;
        const uint64_t rec = fold(raw[i], &isReverse);
        const uint64_t column = ((rec >> 14) & 0x3ffffffull) * 12288ull + (rec & 0x3fffull);
        const uint32_t row = (uint32_t)(rec >> 40);
        size_t j = std::upper_bound(recordStarts_.begin(), recordStarts_.end(), column) - recordStarts_.begin();
        size_t k = std::upper_bound(modelStarts_.begin(), modelStarts_.end(), row) - modelStarts_.begin();
        if (j == 0 || k == 0) return false;
        j--; k--;
        if (column - recordStarts_[j] >= recordLengths_[j]) return false;                     // separator or padding column
        if (row - modelStarts_[k] >= models.lengths[k]) return false;                         // separator row
        *hit = HavacHit(column - recordStarts_[j], (uint32_t)j, row - modelStarts_[k], (uint32_t)k);
        mirror(*hit, isReverse);
        return true;
    });
}

vector<HavacWindow> havacMergeHitsToWindows(const vector<HavacHit> &hits, const vector<uint32_t> &modelLengths,
                                            const vector<uint64_t> &recordLengths, uint32_t flank) {
    vector<HavacWindow> stretches;
    stretches.reserve(hits.size());
    for (const HavacHit &h : hits) {
        if (h.phmmIndex >= modelLengths.size() || h.sequenceIndex >= recordLengths.size()) continue;
        const uint64_t n = recordLengths[h.sequenceIndex];
        const uint64_t length = modelLengths[h.phmmIndex];
        if (n == 0 || length == 0) continue;
        const uint64_t i = std::min<uint64_t>(h.sequencePosition, n - 1);   // a hit on the terminator column
        const uint64_t k = std::min<uint64_t>(h.phmmPosition, length - 1);
        const uint64_t before = (h.reverseStrand ? length - 1 - k : k) + flank;
        const uint64_t after = (h.reverseStrand ? k : length - 1 - k) + flank;
        HavacWindow w;
        w.sequenceIndex = h.sequenceIndex;
        w.phmmIndex = h.phmmIndex;
        w.reverseStrand = h.reverseStrand;
        w.sequenceStart = i > before ? i - before : 0;
        w.sequenceEnd = std::min<uint64_t>(i + after, n - 1);
        w.phmmFirst = w.phmmLast = (uint32_t)k;
        w.hitCount = 1;
        stretches.push_back(w);
    }
    std::sort(stretches.begin(), stretches.end(), [](const HavacWindow &a, const HavacWindow &b) {
        if (a.sequenceIndex != b.sequenceIndex) return a.sequenceIndex < b.sequenceIndex;
        if (a.reverseStrand != b.reverseStrand) return b.reverseStrand;
        if (a.phmmIndex != b.phmmIndex) return a.phmmIndex < b.phmmIndex;
        if (a.sequenceStart != b.sequenceStart) return a.sequenceStart < b.sequenceStart;
        return a.sequenceEnd < b.sequenceEnd;
    });
    vector<HavacWindow> out;
    for (const HavacWindow &w : stretches) {
        if (!out.empty()) {
            HavacWindow &last = out.back();
            if (last.sequenceIndex == w.sequenceIndex && last.reverseStrand == w.reverseStrand &&
                last.phmmIndex == w.phmmIndex && w.sequenceStart <= last.sequenceEnd + 1) {
                last.sequenceEnd = std::max(last.sequenceEnd, w.sequenceEnd);
                last.phmmFirst = std::min(last.phmmFirst, w.phmmFirst);
                last.phmmLast = std::max(last.phmmLast, w.phmmLast);
                last.hitCount += 1;
                continue;
            }
        }
        out.push_back(w);
    }
    return out;
}

vector<HavacWindow> Havac::getWindowsFromFinishedRun(uint32_t flank) {
    vector<uint32_t> modelLengths;
    vector<HavacHit> hits = fetchHits(&modelLengths);
    vector<uint64_t> recordLengths;
    size_t start = 0;
    for (size_t i = 0; i < fastaVector->metadata.count; i++) {          // end position is one past the terminator
        const size_t end = fastaVector->metadata.data[i].sequenceEndPosition;
        recordLengths.push_back(end > start ? end - start - 1 : 0);
        start = end;
    }
    return havacMergeHitsToWindows(hits, modelLengths, recordLengths, flank);
}

HavacHit::HavacHit(const uint64_t sequencePosition, const uint32_t sequenceIndex, const uint32_t phmmPosition,
                   const uint32_t phmmIndex)
    : sequencePosition(sequencePosition), sequenceIndex(sequenceIndex), phmmPosition(phmmPosition), phmmIndex(phmmIndex) {}

std::string HavacHit::toString() {   // host/Havac.cpp:201-206, same text
    std::stringstream ss;
    ss << "sequence $" << sequenceIndex << ", position " << sequencePosition << "; phmm #" << phmmIndex << " position "
       << phmmPosition;
    if (reverseStrand) ss << " (reverse strand)";
    return ss.str();
}
