// havac_pipe.hip -- passes in flight, behind the C ABI (include/havac_dev.h, "Level 2b").
//
// The reference runs one pass at a time: invokeHavacSsvAsync, waitForHavacSsvAsync, getHitList (host/HavacHwClient.cpp:141-157,
// 172-202).  On an MI355X a pass of C2 is 1.85 ms of kernel and ~0.08 ms of everything else (the preparation in front of the
// kernel, the ordering of the records behind it, the host's turnaround), and a kernel's last round of tiles leaves the chip
// half empty: with TWO passes in flight -- each with its own context and hit buffer, consecutive passes alternating between two
// high-priority streams, a whole pass (preparation, SSV kernel, ordering) on ONE of them -- the ordering, the gather (N > 1) and
// the host's work of pass k run beside the kernel of pass k + 1, which itself starts while kernel k drains.  (Rounds 3-4 gave every
// slot a low-priority stream for its ordering, beside ONE kernel stream for short passes: with a pass's launches down from ten
// to eight and no host round trip inside it, two plain streams are faster at every model height -- 64 rows x 100 Mbp 38.4 ->
// 42.2 TCUPS, 256 rows 50.3 -> 53.9, 512 rows 53.0 -> 56.1, C2 unchanged: profiles/r05n_*.)  Rounds 1-4 had this
// engine in Python over torch streams (havac_amd/dist.py: ShardedSsv); a C++ caller of the drop-in API got the strictly serial
// figure.  Since round 5 a pass is enqueued whole (three launches for the kernel, five for the ordering, no
// host round trip: havac_dev.hip, hit_order.hip.h), so the engine is a few lines of stream bookkeeping -- here, in C++, used
// by the handle API (havac_dev_set_pipeline_depth), by the `Havac` class and, through ctypes, by dist.py and bench.py.
#include "../../include/havac_dev.h"

#include <algorithm>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace {


struct Slot {
    havac_ssv_ctx* ctx = nullptr;
    uint64_t* d_hits = nullptr;
    hipStream_t stream = nullptr;              // the stream the slot's pass in flight was submitted on
    uint64_t* merged = nullptr; uint64_t merged_capacity = 0;      // rank 0 of a sharded run: the gather's receive buffer
    hipEvent_t gathered = nullptr;             // behind the slot's last gather: its hit buffer may be written again
    hipEvent_t g0 = nullptr, g1 = nullptr;     // timing of the slot's last gather
    bool timed = false;
};

}  // namespace

struct havac_pipe {
    int device = 0;
    uint32_t depth = 1;
    uint64_t hit_capacity = 0;
    int kernel_streams = 1;                    // 1: every pass on one stream; 2 .. 4: consecutive passes take the streams in turn
    hipStream_t kstream[4] = {nullptr, nullptr, nullptr, nullptr};
    int flip = 0;
    int streams_used = 1;                      // how many streams the last submit took its turn over
    bool used_two_streams = false;
    std::vector<Slot> slots;
    std::deque<uint32_t> in_flight;            // slot numbers, oldest first
    uint32_t next = 0, last = 0;               // next slot to submit; slot of the last collected pass
    havac_gather* gather = nullptr; uint32_t rank = 0, world = 1;
    hipEvent_t inputs = nullptr;               // the caller's stream at submit time
    std::vector<float> gather_ms;
    const uint64_t* last_records = nullptr;    // what the last collect returned (kept by release)
    bool released = false;
    std::string err;
};

#define PIPE_HIP(p, expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (p)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return _e == hipErrorOutOfMemory ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME;                 \
        }                                                                                       \
    } while (0)

static void free_slot(Slot& s, const uint64_t* keep) {
    if (s.ctx) { havac_ssv_ctx_destroy(s.ctx); s.ctx = nullptr; }
    if (s.d_hits && s.d_hits != keep) { (void)hipFree(s.d_hits); s.d_hits = nullptr; }
    if (s.merged && s.merged != keep) { (void)hipFree(s.merged); s.merged = nullptr; s.merged_capacity = 0; }
}

extern "C" void havac_pipe_destroy(havac_pipe* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (int k = 0; k < 4; k++) if (p->kstream[k]) (void)hipStreamSynchronize(p->kstream[k]);
    for (Slot& s : p->slots) {
        free_slot(s, nullptr);
        if (s.gathered) (void)hipEventDestroy(s.gathered);
        if (s.g0) (void)hipEventDestroy(s.g0);
        if (s.g1) (void)hipEventDestroy(s.g1);
    }
    if (p->inputs) (void)hipEventDestroy(p->inputs);
    for (int k = 0; k < 4; k++) if (p->kstream[k]) (void)hipStreamDestroy(p->kstream[k]);
    delete p;
}

extern "C" int havac_pipe_create(uint32_t depth, uint64_t hit_capacity, int kernel_streams, havac_pipe** out) {
    if (!out || depth == 0 || depth > 8 || kernel_streams == 0 || kernel_streams > 4) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HAVAC_E_NO_DEVICE;
    havac_pipe* p = new (std::nothrow) havac_pipe;
    if (!p) return HAVAC_E_NOMEM;
    p->depth = depth; p->hit_capacity = hit_capacity;
    // (-1, the library's rule: two.  Three or four are for experiments: with three, 64 rows x 100 Mbp gained 10 % per step on three
    // boxes (44.8 -> 49.5 TCUPS), 32 rows gained 11 % on one box and LOST 14 % on two, taller passes moved by -9 ... +2 %
    // (profiles/r05t_step_probe_*): how the command processor interleaves three queues of eight short launches each is not
    // something a rule can be built on; two streams have measured the same on every box of rounds 4 and 5.)
    p->kernel_streams = kernel_streams < 0 ? std::min<int>(2, (int)depth) : std::min<int>(kernel_streams, (int)depth);
    auto body = [&]() -> int {
        PIPE_HIP(p, hipGetDevice(&p->device));
        int least = 0, greatest = 0;
        PIPE_HIP(p, hipDeviceGetStreamPriorityRange(&least, &greatest));
        // High priority.  Two streams where passes are in flight: a pass -- its preparation, its SSV kernel, the ordering of its
        // records -- is one stream's business, consecutive passes alternate, so a kernel starts while its predecessor drains (the
        // last, half-empty round of a launch's tiles and the ~16 us between two dependent launches are filled by its neighbour's
        // first workgroups: C2 1.878 -> 1.78 ms per step, less than one kernel takes alone) and runs beside its predecessor's ordering
        for (int k = 0; k < p->kernel_streams; k++) PIPE_HIP(p, hipStreamCreateWithPriority(&p->kstream[k], hipStreamNonBlocking, greatest));
        (void)least;
        PIPE_HIP(p, hipEventCreateWithFlags(&p->inputs, hipEventDisableTiming));
        p->slots.resize(depth);
        for (Slot& s : p->slots) {
            if (havac_ssv_ctx_create(&s.ctx) != HAVAC_OK) { p->err = "could not create an SSV context"; return HAVAC_E_RUNTIME; }
            if (hit_capacity) PIPE_HIP(p, hipMalloc(&s.d_hits, hit_capacity * sizeof(uint64_t)));
            PIPE_HIP(p, hipEventCreateWithFlags(&s.gathered, hipEventDisableTiming));
            PIPE_HIP(p, hipEventCreate(&s.g0));
            PIPE_HIP(p, hipEventCreate(&s.g1));
        }
        return HAVAC_OK;
    };
    const int rc = body();
    if (rc != HAVAC_OK) { havac_pipe_destroy(p); return rc; }
    *out = p;
    return HAVAC_OK;
}

extern "C" const char* havac_pipe_last_error(havac_pipe* p) { return p ? p->err.c_str() : "null pipe"; }
extern "C" uint32_t havac_pipe_depth(havac_pipe* p) { return p ? p->depth : 0; }
extern "C" uint32_t havac_pipe_in_flight(havac_pipe* p) { return p ? (uint32_t)p->in_flight.size() : 0; }
extern "C" int havac_pipe_used_two_streams(havac_pipe* p) { return p && p->used_two_streams ? 1 : 0; }
extern "C" int havac_pipe_streams_used(havac_pipe* p) { return p ? p->streams_used : 0; }

extern "C" havac_ssv_ctx* havac_pipe_context(havac_pipe* p, int which) {
    if (!p || p->released) return nullptr;
    if (which == -2) return p->slots[p->next].ctx;           // of the next submit
    if (which < 0) return p->slots[p->last].ctx;             // of the last collected pass
    return (uint32_t)which < p->depth ? p->slots[which].ctx : nullptr;
}

extern "C" int havac_pipe_set_gather(havac_pipe* p, havac_gather* g) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (!p->in_flight.empty()) { p->err = "passes are in flight: set the gather between passes"; return HAVAC_E_LOGIC; }
    p->gather = g; p->rank = 0; p->world = 1;
    if (g) return havac_gather_info(g, &p->rank, &p->world);
    return HAVAC_OK;
}

extern "C" int havac_pipe_submit(havac_pipe* p, const uint8_t* d_sequence, uint64_t nsymbols, const int8_t* d_phmm, uint32_t nrows,
                                 uint32_t shard_index, uint32_t shard_count, const uint32_t* d_abort_flag, void* caller_stream) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (p->released) { p->err = "the pipe was released"; return HAVAC_E_LOGIC; }
    if (p->in_flight.size() == p->depth) { p->err = "every slot is in flight: collect first"; return HAVAC_E_LOGIC; }
    PIPE_HIP(p, hipSetDevice(p->device));
    Slot& s = p->slots[p->next];
    hipStream_t stream = p->kstream[0];
    if (p->kstream[1]) {
        const int n = p->kernel_streams;
        p->used_two_streams = true;
        p->streams_used = n;
        p->flip = (p->flip + 1) % n;
        stream = p->kstream[p->flip];
    }
    s.stream = stream;
    // the caller's inputs (HAVAC_NO_STREAM: they are in place, nothing to wait for -- an event on the legacy null stream alone
    // costs a strictly serial 0.2 ms pass 5 %); and, in a sharded run, the slot's last gather, which reads the hit buffer this pass writes
    if (caller_stream != HAVAC_NO_STREAM) {
        PIPE_HIP(p, hipEventRecord(p->inputs, (hipStream_t)caller_stream));
        PIPE_HIP(p, hipStreamWaitEvent(stream, p->inputs, 0));
    }
    if (p->gather) PIPE_HIP(p, hipStreamWaitEvent(stream, s.gathered, 0));
    const int rc = havac_ssv_enqueue(s.ctx, d_sequence, nsymbols, d_phmm, nrows, shard_index, shard_count, s.d_hits, p->hit_capacity, d_abort_flag, stream);
    if (rc) { p->err = havac_ssv_ctx_last_error(s.ctx); return rc; }
    p->in_flight.push_back(p->next);
    p->next = (p->next + 1) % p->depth;
    return HAVAC_OK;
}

extern "C" int havac_pipe_poll(havac_pipe* p) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (p->in_flight.empty()) { p->err = "nothing in flight"; return HAVAC_E_LOGIC; }
    const int rc = havac_ssv_query(p->slots[p->in_flight.front()].ctx);
    if (rc < 0) p->err = havac_ssv_ctx_last_error(p->slots[p->in_flight.front()].ctx);
    return rc;
}

extern "C" int havac_pipe_wait_inputs(havac_pipe* p, int sequence_too) {
    if (!p) return HAVAC_E_ARGUMENT;
    for (uint32_t slot : p->in_flight)
        if (int rc = havac_ssv_wait_inputs(p->slots[slot].ctx, sequence_too)) { p->err = havac_ssv_ctx_last_error(p->slots[slot].ctx); return rc; }
    return HAVAC_OK;
}

static void harvest(havac_pipe* p, Slot& s) {
    if (!s.timed) return;
    float ms = 0.f;
    if (hipEventSynchronize(s.g1) == hipSuccess && hipEventElapsedTime(&ms, s.g0, s.g1) == hipSuccess) p->gather_ms.push_back(ms);
    s.timed = false;
}

extern "C" int havac_pipe_collect(havac_pipe* p, uint64_t* found_out, const uint64_t** d_records_out, uint64_t* nrecords_out, void* caller_stream) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (p->in_flight.empty()) { p->err = "nothing in flight"; return HAVAC_E_LOGIC; }
    PIPE_HIP(p, hipSetDevice(p->device));
    const uint32_t slot = p->in_flight.front();
    p->in_flight.pop_front();
    p->last = slot;
    Slot& s = p->slots[slot];
    uint64_t found = 0;
    const int pass_rc = havac_ssv_finish(s.ctx, &found);
    if (pass_rc) p->err = havac_ssv_ctx_last_error(s.ctx);
    if (found_out) *found_out = found;
    if (d_records_out) *d_records_out = nullptr;
    if (nrecords_out) *nrecords_out = 0;
    if (!p->gather) {
        if (pass_rc) return pass_rc;
        p->last_records = s.d_hits;
        if (d_records_out) *d_records_out = s.d_hits;
        if (nrecords_out) *nrecords_out = found;
        return HAVAC_OK;
    }
    // A sharded run: the count exchange still runs on every rank when this rank's pass failed (-1), and every rank returns an
    // error afterwards, so no rank is left waiting inside a collective.
    if (caller_stream == HAVAC_NO_STREAM) caller_stream = nullptr;
    const hipStream_t stream = s.stream ? s.stream : (hipStream_t)caller_stream;      // (idle: the pass has been waited for; the next pass of this slot queues behind)
    harvest(p, s);
    PIPE_HIP(p, hipEventRecord(s.g0, stream));
    std::vector<int64_t> counts(p->world, 0);
    int rc = havac_gather_counts(p->gather, pass_rc ? -1 : (int64_t)found, counts.data(), stream);
    if (rc) { p->err = havac_gather_last_error(p->gather); return rc; }
    uint64_t total = 0;
    std::string failed;
    for (uint32_t r = 0; r < p->world; r++) {
        if (counts[r] < 0) failed += (failed.empty() ? "" : ", ") + std::to_string(r);
        else total += (uint64_t)counts[r];
    }
    if (!failed.empty()) {
        if (pass_rc) return pass_rc;                        // this rank's own error says more
        p->err = "the pass failed on rank(s) " + failed + "; no records were gathered";
        return HAVAC_E_RUNTIME;
    }
    if (p->rank == 0 && s.merged_capacity < total) {
        PIPE_HIP(p, hipStreamSynchronize(stream));              // (the buffer's last gather)
        if (s.merged) { if (s.merged == p->last_records) p->last_records = nullptr; (void)hipFree(s.merged); }      // (its records were valid until this slot was submitted again)
        s.merged = nullptr; s.merged_capacity = 0;
        PIPE_HIP(p, hipMalloc(&s.merged, std::max<uint64_t>(total, 1) * sizeof(uint64_t)));
        s.merged_capacity = std::max<uint64_t>(total, 1);
    }
    rc = havac_gather_records(p->gather, s.d_hits, p->rank == 0 ? s.merged : nullptr, s.merged_capacity, stream);
    if (rc) { p->err = havac_gather_last_error(p->gather); return rc; }
    PIPE_HIP(p, hipEventRecord(s.g1, stream));
    s.timed = true;
    PIPE_HIP(p, hipEventRecord(s.gathered, stream));
    if (stream != (hipStream_t)caller_stream) PIPE_HIP(p, hipStreamWaitEvent((hipStream_t)caller_stream, s.gathered, 0));      // the records are the caller's
    if (p->rank == 0) {
        p->last_records = s.merged;
        if (d_records_out) *d_records_out = s.merged;
        if (nrecords_out) *nrecords_out = total;
    }
    return HAVAC_OK;
}

// `nsteps` passes of the same inputs, as many in flight as the pipe is deep, all complete on return: the loop a caller makes --
// submit, and collect the oldest once every slot is in flight -- without the caller's language in it (bench.py's timed regions:
// what is measured is then what a C++ caller of this library gets).  kernel_ms / total_ms (nsteps entries each, may be NULL)
// receive every pass's havac_pipe_last_ms; the last pass's result as havac_pipe_collect returns it.
extern "C" int havac_pipe_run(havac_pipe* p, uint32_t nsteps, const uint8_t* d_sequence, uint64_t nsymbols, const int8_t* d_phmm, uint32_t nrows,
                              uint32_t shard_index, uint32_t shard_count, void* caller_stream, float* kernel_ms, float* total_ms,
                              uint64_t* found_out, const uint64_t** d_records_out, uint64_t* nrecords_out) {
    if (!p) return HAVAC_E_ARGUMENT;
    uint32_t collected = 0;
    auto collect = [&]() -> int {
        const int rc = havac_pipe_collect(p, found_out, d_records_out, nrecords_out, caller_stream);
        if (rc) return rc;
        float k = 0.f, t = 0.f;
        (void)havac_pipe_last_ms(p, &k, &t);
        if (kernel_ms) kernel_ms[collected] = k;
        if (total_ms) total_ms[collected] = t;
        collected++;
        return HAVAC_OK;
    };
    for (uint32_t i = 0; i < nsteps; i++) {
        if (int rc = havac_pipe_submit(p, d_sequence, nsymbols, d_phmm, nrows, shard_index, shard_count, nullptr, i == 0 ? caller_stream : HAVAC_NO_STREAM)) return rc;
        if (p->in_flight.size() == p->depth) { if (int rc = collect()) return rc; }
    }
    while (!p->in_flight.empty()) { if (int rc = collect()) return rc; }
    return HAVAC_OK;
}

extern "C" int havac_pipe_wait_gathers(havac_pipe* p) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (!p->gather) return HAVAC_OK;
    const int rc = havac_gather_wait(p->gather);
    if (rc) p->err = havac_gather_last_error(p->gather);
    return rc;
}

extern "C" int havac_pipe_gather_times(havac_pipe* p, float* out_ms, uint32_t capacity, uint32_t* count) {
    if (!p || !count) return HAVAC_E_ARGUMENT;
    for (Slot& s : p->slots) harvest(p, s);
    *count = (uint32_t)p->gather_ms.size();
    for (uint32_t i = 0; i < *count && i < capacity && out_ms; i++) out_ms[i] = p->gather_ms[i];
    p->gather_ms.clear();
    return HAVAC_OK;
}

extern "C" int havac_pipe_last_ms(havac_pipe* p, float* ssv_kernel_ms, float* total_ms) {
    if (!p || p->released) return HAVAC_E_ARGUMENT;
    return havac_ssv_last_ms(p->slots[p->last].ctx, ssv_kernel_ms, total_ms);
}

extern "C" int havac_pipe_release(havac_pipe* p) {
    if (!p) return HAVAC_E_ARGUMENT;
    if (!p->in_flight.empty()) { p->err = "passes are in flight: collect them first"; return HAVAC_E_LOGIC; }
    PIPE_HIP(p, hipSetDevice(p->device));
    for (int k = 0; k < 4; k++) if (p->kstream[k]) PIPE_HIP(p, hipStreamSynchronize(p->kstream[k]));
    for (Slot& s : p->slots) {
        harvest(p, s);
        free_slot(s, p->last_records);
    }
    p->released = true;
    return HAVAC_OK;
}
