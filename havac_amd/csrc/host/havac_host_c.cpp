// havac_host_c.cpp -- include/havac_host.h: C wrappers around class Havac and the host-only stages.
#include "../../../include/havac_host.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

#include "../../../include/havac_dev.h"
#include "Havac.hpp"
#include "PhmmPreprocessor.hpp"
#include "PhmmReprojection.h"
#include "SequencePreprocessor.hpp"

struct havac_host {
    Havac *obj = nullptr;
    std::string err;
    vector<HavacHit> hits;
    bool haveHits = false;
    uint32_t depth = 1;
};

namespace {
template <typename F>
int guarded(havac_host *h, F &&f) {
    try {
        f();
        return HAVAC_OK;
    } catch (const std::length_error &e) { if (h) h->err = e.what(); return HAVAC_E_LENGTH;
    } catch (const std::overflow_error &e) { if (h) h->err = e.what(); return HAVAC_E_HIT_OVERFLOW;
    } catch (const std::logic_error &e) { if (h) h->err = e.what(); return HAVAC_E_LOGIC;
    } catch (const std::bad_alloc &e) { if (h) h->err = e.what(); return HAVAC_E_NOMEM;
    } catch (const std::exception &e) { if (h) h->err = e.what(); return HAVAC_E_RUNTIME; }
}

int copyHits(const vector<HavacHit> &hits, uint64_t *sp, uint32_t *si, uint32_t *pp, uint32_t *pi, uint32_t cap,
             uint32_t *count) {
    if (hits.size() > 0xffffffffull) return HAVAC_E_HIT_OVERFLOW;   // these FFI wrappers count in 32 bits; the C++ class does not
    if (count) *count = (uint32_t)hits.size();
    for (uint32_t i = 0; i < cap && i < hits.size(); i++) {
        if (sp) sp[i] = hits[i].sequencePosition;
        if (si) si[i] = hits[i].sequenceIndex;
        if (pp) pp[i] = hits[i].phmmPosition;
        if (pi) pi[i] = hits[i].phmmIndex;
    }
    return HAVAC_OK;
}
}  // namespace

extern "C" {

int havac_host_create(uint32_t device_index, float p, havac_host **out) {
    if (!out) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    havac_host *h = new (std::nothrow) havac_host;
    if (!h) return HAVAC_E_NOMEM;
    int rc = guarded(h, [&] { h->obj = new Havac(device_index, p); });
    if (rc != HAVAC_OK) { delete h; return rc == HAVAC_E_RUNTIME ? HAVAC_E_NO_DEVICE : rc; }
    *out = h;
    return HAVAC_OK;
}

int havac_host_create_deferred(uint32_t device_index, float p, havac_host **out) {
    if (!out) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    havac_host *h = new (std::nothrow) havac_host;
    if (!h) return HAVAC_E_NOMEM;
    int rc = guarded(h, [&] { h->obj = new Havac(Havac::DeferredStart{}, device_index, p); });
    if (rc != HAVAC_OK) { delete h; return rc; }
    *out = h;
    return HAVAC_OK;
}

int havac_host_create_multi(const uint32_t *devices, uint32_t n, float p, havac_host **out) {
    if (!out || !devices || n == 0) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    havac_host *h = new (std::nothrow) havac_host;
    if (!h) return HAVAC_E_NOMEM;
    int rc = guarded(h, [&] { h->obj = new Havac(std::vector<uint32_t>(devices, devices + n), p); });
    if (rc != HAVAC_OK) { delete h; return rc == HAVAC_E_RUNTIME ? HAVAC_E_NO_DEVICE : rc; }
    *out = h;
    return HAVAC_OK;
}

void havac_host_destroy(havac_host *h) {
    if (!h) return;
    delete h->obj;
    delete h;
}

const char *havac_host_last_error(havac_host *h) { return h ? h->err.c_str() : "null handle"; }

int havac_host_load_sequence(havac_host *h, const char *p) { return guarded(h, [&] { h->obj->loadSequence(p); }); }
int havac_host_load_phmm(havac_host *h, const char *p) { return guarded(h, [&] { h->obj->loadPhmm(p); }); }
int havac_host_run(havac_host *h) { h->haveHits = false; return guarded(h, [&] { h->obj->runHardwareClient(); }); }
int havac_host_run_async(havac_host *h) { h->haveHits = false; return guarded(h, [&] { h->obj->runHardwareClientAsync(); }); }
int havac_host_wait(havac_host *h) { return guarded(h, [&] { h->obj->waitHardwareClientAsync(); }); }
int havac_host_abort(havac_host *h) { return guarded(h, [&] { h->obj->abortHardwareClient(); }); }
int havac_host_set_hit_capacity(havac_host *h, uint64_t n) { return guarded(h, [&] { h->obj->setHitCapacity(n); }); }
int havac_host_set_pipeline_depth(havac_host *h, uint32_t depth) {
    const int rc = guarded(h, [&] { h->obj->setPipelineDepth(depth); });
    if (rc == HAVAC_OK) { h->depth = depth; h->haveHits = false; }
    return rc;
}
// With several runs open havac_host_get_hits fetches (and closes) the oldest run once and then serves the copy it keeps -- a
// caller asks twice, for the count and for the arrays; this says the caller is done with that run: the next havac_host_get_hits
// fetches the next one.  At depth 1 the copy is kept until the next run, as the reference's getHitsFromFinishedRun can be
// called again and again.
int havac_host_next_run(havac_host *h) {
    if (h->depth > 1) h->haveHits = false;
    return HAVAC_OK;
}
int havac_host_set_both_strands(havac_host *h, int on) { return guarded(h, [&] { h->obj->setBothStrands(on != 0); }); }
int havac_host_get_hit_strands(havac_host *h, uint8_t *reverse, uint32_t cap, uint32_t *count) {
    if (count) *count = (uint32_t)h->hits.size();
    for (uint32_t i = 0; i < cap && i < h->hits.size(); i++) reverse[i] = h->hits[i].reverseStrand ? 1 : 0;
    return HAVAC_OK;
}
int havac_host_set_device_packing(havac_host *h, int on) { return guarded(h, [&] { h->obj->setDevicePacking(on != 0); }); }
int havac_host_set_boundary_mode(havac_host *h, int on) { return guarded(h, [&] { h->obj->setBoundaryMode(on != 0); }); }

int havac_host_state(havac_host *h) {
    int s = 0;
    int rc = guarded(h, [&] { s = (int)h->obj->currentHardwareState(); });
    return rc == HAVAC_OK ? s : rc;
}

int havac_host_get_hits(havac_host *h, uint64_t *sp, uint32_t *si, uint32_t *pp, uint32_t *pi, uint32_t cap,
                        uint32_t *count) {
    if (!h->haveHits) {
        int rc = guarded(h, [&] { h->hits = h->obj->getHitsFromFinishedRun(); });
        if (rc != HAVAC_OK) return rc;
        h->haveHits = true;
    }
    return copyHits(h->hits, sp, si, pp, pi, cap, count);
}

static int copyWindows(const vector<HavacWindow> &w, uint32_t *si, uint32_t *pi, uint8_t *rs, uint64_t *start,
                       uint64_t *end, uint32_t *pf, uint32_t *pl, uint32_t *hc, uint32_t cap, uint32_t *count) {
    if (count) *count = (uint32_t)w.size();
    for (uint32_t i = 0; i < cap && i < w.size(); i++) {
        si[i] = w[i].sequenceIndex; pi[i] = w[i].phmmIndex; rs[i] = w[i].reverseStrand ? 1 : 0;
        start[i] = w[i].sequenceStart; end[i] = w[i].sequenceEnd;
        pf[i] = w[i].phmmFirst; pl[i] = w[i].phmmLast; hc[i] = w[i].hitCount;
    }
    return HAVAC_OK;
}

int havac_host_get_windows(havac_host *h, uint32_t flank, uint32_t *si, uint32_t *pi, uint8_t *rs, uint64_t *start,
                           uint64_t *end, uint32_t *pf, uint32_t *pl, uint32_t *hc, uint32_t cap, uint32_t *count) {
    vector<HavacWindow> w;
    int rc = guarded(h, [&] { w = h->obj->getWindowsFromFinishedRun(flank); });
    if (rc != HAVAC_OK) return rc;
    return copyWindows(w, si, pi, rs, start, end, pf, pl, hc, cap, count);
}

int havac_host_merge_windows(const uint64_t *sp, const uint32_t *sidx, const uint32_t *pp, const uint32_t *pidx,
                             const uint8_t *reverse, uint32_t nhits, const uint32_t *modelLengths, uint32_t nmodels,
                             const uint64_t *recordLengths, uint32_t nrecords, uint32_t flank, uint32_t *si,
                             uint32_t *pi, uint8_t *rs, uint64_t *start, uint64_t *end, uint32_t *pf, uint32_t *pl,
                             uint32_t *hc, uint32_t cap, uint32_t *count) {
    try {
        vector<HavacHit> hits;
        hits.reserve(nhits);
        for (uint32_t i = 0; i < nhits; i++) {
            HavacHit hit(sp[i], sidx[i], pp[i], pidx[i]);
            hit.reverseStrand = reverse && reverse[i];
            hits.push_back(hit);
        }
        vector<HavacWindow> w = havacMergeHitsToWindows(hits, vector<uint32_t>(modelLengths, modelLengths + nmodels),
                                                        vector<uint64_t>(recordLengths, recordLengths + nrecords), flank);
        return copyWindows(w, si, pi, rs, start, end, pf, pl, hc, cap, count);
    } catch (const std::bad_alloc &) {
        return HAVAC_E_NOMEM;
    }
}

int havac_host_get_raw_hits(havac_host *h, uint64_t *out, uint32_t cap, uint32_t *count) {
    const vector<uint64_t> &raw = h->obj->rawHitsOfLastFetch();
    if (count) *count = (uint32_t)raw.size();
    for (uint32_t i = 0; i < cap && i < raw.size(); i++) out[i] = raw[i];
    return HAVAC_OK;
}

int havac_host_last_run_ms(havac_host *h, float *a, float *b) {
    return guarded(h, [&] { h->obj->lastRunMilliseconds(a, b); });
}

// ---- host-only stages ------------------------------------------------------

int havac_host_pack_fasta(const char *path, int64_t seed, uint8_t *out, uint64_t cap, uint64_t *nbytes,
                          uint64_t *nchars, uint32_t *nrecords) {
    FastaVector fv;
    if (fastaVectorInit(&fv) != FASTA_VECTOR_OK) return HAVAC_E_NOMEM;
    FastaVectorReturnCode rc = fastaVectorReadFasta(path, &fv);
    if (rc != FASTA_VECTOR_OK) { fastaVectorDealloc(&fv); return rc == FASTA_VECTOR_ALLOCATION_FAIL ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME; }
    if (seed >= 0) std::srand((unsigned)seed);
    SequencePreprocessor pre(&fv);
    vector<uint8_t> &packed = pre.getCompressedSequenceBuffer();
    if (nbytes) *nbytes = packed.size();
    if (nchars) *nchars = fv.sequence.count;
    if (nrecords) *nrecords = (uint32_t)fv.metadata.count;
    if (out && cap >= packed.size()) std::memcpy(out, packed.data(), packed.size());
    fastaVectorDealloc(&fv);
    return HAVAC_OK;
}

int havac_host_pack_fasta_layout(const char *path, int64_t seed, int boundaryMode, int bothStrands, uint8_t *out, uint64_t cap,
                                 uint64_t *nbytes, uint8_t *maskOut, uint64_t maskCap, uint64_t *maskBytes,
                                 uint64_t *forwardColumns) {
    FastaVector fv;
    if (fastaVectorInit(&fv) != FASTA_VECTOR_OK) return HAVAC_E_NOMEM;
    FastaVectorReturnCode rc = fastaVectorReadFasta(path, &fv);
    if (rc != FASTA_VECTOR_OK) { fastaVectorDealloc(&fv); return rc == FASTA_VECTOR_ALLOCATION_FAIL ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME; }
    if (seed >= 0) std::srand((unsigned)seed);
    SequencePreprocessor pre(&fv, boundaryMode != 0);
    uint64_t nf = 0;
    if (bothStrands) {
        vector<uint64_t> starts, residues;
        for (size_t j = 0; j < fv.metadata.count; j++) {
            const uint64_t begin = j ? fv.metadata.data[j - 1].sequenceEndPosition : 0;
            starts.push_back(boundaryMode ? pre.getRecordStarts()[j] : begin);
            residues.push_back(fv.metadata.data[j].sequenceEndPosition - begin - 1);
        }
        nf = pre.appendReverseStrand(starts, residues);
    }
    vector<uint8_t> &packed = pre.getCompressedSequenceBuffer();
    vector<uint8_t> &mask = pre.getSeparatorMask();
    if (nbytes) *nbytes = packed.size();
    if (maskBytes) *maskBytes = mask.size();
    if (forwardColumns) *forwardColumns = nf;
    if (out && cap >= packed.size()) std::memcpy(out, packed.data(), packed.size());
    if (maskOut && maskCap >= mask.size() && !mask.empty()) std::memcpy(maskOut, mask.data(), mask.size());
    fastaVectorDealloc(&fv);
    return HAVAC_OK;
}

int havac_host_text_and_patches(const char *path, int64_t seed, char *chars, uint64_t charsCap, uint64_t *nchars,
                                uint64_t *patchColumns, uint8_t *patchSymbols, uint64_t patchCap, uint64_t *npatches) {
    FastaVector fv;
    if (fastaVectorInit(&fv) != FASTA_VECTOR_OK) return HAVAC_E_NOMEM;
    FastaVectorReturnCode rc = fastaVectorReadFasta(path, &fv);
    if (rc != FASTA_VECTOR_OK) { fastaVectorDealloc(&fv); return rc == FASTA_VECTOR_ALLOCATION_FAIL ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME; }
    if (seed >= 0) std::srand((unsigned)seed);
    vector<uint64_t> columns;
    vector<uint8_t> symbols;
    SequencePreprocessor::collectPatches(&fv, columns, symbols);
    if (nchars) *nchars = fv.sequence.count;
    if (npatches) *npatches = columns.size();
    if (chars && charsCap >= fv.sequence.count) std::memcpy(chars, fv.sequence.charData, fv.sequence.count);
    if (patchColumns && patchSymbols && patchCap >= columns.size()) {
        std::memcpy(patchColumns, columns.data(), columns.size() * sizeof(uint64_t));
        std::memcpy(patchSymbols, symbols.data(), symbols.size());
    }
    fastaVectorDealloc(&fv);
    return HAVAC_OK;
}

int havac_host_project_hmm(const char *path, float p, int8_t *out, uint64_t cap, uint64_t *nbytes, uint32_t *nmodels,
                           uint32_t *lengths, uint32_t lengths_cap) {
    P7HmmList list;
    P7HmmReturnCode rc = readP7Hmm(path, &list);
    if (rc == p7HmmAllocationFailure) return HAVAC_E_NOMEM;
    if (rc != p7HmmSuccess) return HAVAC_E_RUNTIME;
    std::shared_ptr<std::vector<int8_t>> data;
    try {
        data = PhmmPreprocessor(&list, p).getProcessedPhmmData();
    } catch (const std::bad_alloc &) {
        p7HmmListDealloc(&list);
        return HAVAC_E_NOMEM;
    } catch (const std::exception &) {          // a model that is not a nucleotide model
        p7HmmListDealloc(&list);
        return HAVAC_E_ARGUMENT;
    }
    if (nbytes) *nbytes = data->size();
    if (nmodels) *nmodels = list.count;
    for (uint32_t i = 0; lengths && i < list.count && i < lengths_cap; i++) lengths[i] = list.phmms[i].header.modelLength;
    if (out && cap >= data->size()) std::memcpy(out, data->data(), data->size());
    p7HmmListDealloc(&list);
    return HAVAC_OK;
}

int havac_host_read_hmm_emissions(const char *path, float *out, uint64_t cap, uint64_t *nvalues, uint32_t *nmodels) {
    P7HmmList list;
    P7HmmReturnCode rc = readP7Hmm(path, &list);
    if (rc == p7HmmAllocationFailure) return HAVAC_E_NOMEM;
    if (rc != p7HmmSuccess) return HAVAC_E_RUNTIME;
    uint64_t at = 0;
    for (uint32_t i = 0; i < list.count; i++) {
        const uint64_t n = (uint64_t)list.phmms[i].header.modelLength * p7HmmGetAlphabetCardinality(&list.phmms[i]);
        for (uint64_t k = 0; k < n; k++, at++)
            if (out && at < cap) out[at] = list.phmms[i].model.matchEmissionScores[k];
    }
    if (nvalues) *nvalues = at;
    if (nmodels) *nmodels = list.count;
    p7HmmListDealloc(&list);
    return HAVAC_OK;
}

float havac_host_scaling_factor(float mu, float lambda, uint32_t max_length, uint32_t model_length, float p) {
    P7Hmm h;
    std::memset(&h, 0, sizeof h);
    h.stats.msvGumbelMu = mu; h.stats.msvGumbelLambda = lambda;
    h.header.maxLength = max_length; h.header.modelLength = model_length;
    return findThreshold256ScalingFactor(&h, p);
}

int havac_host_project_model(float mu, float lambda, uint32_t max_length, uint32_t model_length, float p,
                             const float *emissions, int8_t *out) {
    if (!emissions || !out || model_length == 0) return HAVAC_E_ARGUMENT;
    P7Hmm h;
    std::memset(&h, 0, sizeof h);
    h.header.alphabet = P7HmmReaderAlphabetDna;
    h.stats.msvGumbelMu = mu; h.stats.msvGumbelLambda = lambda;
    h.header.maxLength = max_length; h.header.modelLength = model_length;
    h.model.matchEmissionScores = const_cast<float *>(emissions);
    p7HmmProjectForThreshold256(&h, p, out);
    return HAVAC_OK;
}

double havac_host_gumbel_invsurv(double p, double mu, double lambda) { return esl_gumbel_invsurv(p, mu, lambda); }

float havac_host_project_score(float s, float m) { return emissionScoreToProjectedScore(s, m); }

int havac_host_resolve_hits(const char *fasta, const char *hmm, const uint64_t *raw, uint32_t nraw, uint64_t *sp,
                            uint32_t *si, uint32_t *pp, uint32_t *pi, uint32_t cap, uint32_t *count) {
    FastaVector fv;
    if (fastaVectorInit(&fv) != FASTA_VECTOR_OK) return HAVAC_E_NOMEM;
    if (fastaVectorReadFasta(fasta, &fv) != FASTA_VECTOR_OK) { fastaVectorDealloc(&fv); return HAVAC_E_RUNTIME; }
    P7HmmList list;
    if (readP7Hmm(hmm, &list) != p7HmmSuccess) { fastaVectorDealloc(&fv); return HAVAC_E_RUNTIME; }
    vector<uint32_t> sums(1, 0u);
    for (uint32_t i = 0; i < list.count; i++) sums.push_back(sums.back() + list.phmms[i].header.modelLength);
    vector<uint64_t> r(raw, raw + nraw);
    vector<HavacHit> hits = havacResolveHits(r, &fv, sums);
    copyHits(hits, sp, si, pp, pi, cap, count);
    p7HmmListDealloc(&list);
    fastaVectorDealloc(&fv);
    return HAVAC_OK;
}

}  // extern "C"
