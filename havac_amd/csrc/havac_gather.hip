// havac_gather.hip -- the one exchange of a sharded run, behind the C ABI (include/havac_dev.h, "Level 3").
//
// north_star: "a thin C-ABI HIP layer ... the sequence x model search space shards embarrassingly across the 8 GPUs of one
// node with only an RCCL gather of HavacHit records over xGMI at the end".  The reference has no counterpart: one
// deviceIndex per object (host/Havac.hpp:51).  One process per GPU; every rank orders its own records (havac_ssv_finish),
// and because shards are runs of whole 12288-column segments the ranks' lists laid end to end in rank order ARE the
// reference's device order (device/HavacHls.cpp:151-152,264): rank 0 receives, it never sorts.
//
// The exchange is variable-length and exact:
//   1. ncclAllGather of one int64 per rank: its record count (-1: this rank's pass failed);
//   2. one ncclGroup: every rank r > 0 ncclSend()s exactly count[r] records, rank 0 ncclRecv()s each list straight into
//      ONE buffer at the exclusive-scan offset of its rank (its own list: one device-to-device copy).
// Nothing is padded to the longest list, nothing is concatenated afterwards: rank 0's extra memory is sum(count) * 8 bytes
// (C4: 4.46e9 records = 36 GB).  xGMI is point-to-point (7 links per GPU): seven senders reach rank 0 over seven different
// links, so the gather is bound by rank 0's HBM write rate and its links, not by a ring.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy that is already in the process if there is one -- PyTorch
// ships its own -- else the system's), so libhavac_dev.so keeps its two dependencies (HIP, the C++ runtime) and a
// single-GPU caller never loads a collective library.  No torch type, no MPI: the 128-byte id travels over whatever side
// channel the caller has (a file, a socket, torch.distributed's store).
#include "../../include/havac_dev.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// The library to bind: librccl.so.1 unless havac_gather_use_library() named another one before the first use (rehearsals of
// several ranks on ONE GPU, which RCCL itself refuses: tests/native/rccl_standin.cpp).  An argument, not an environment variable.
std::mutex g_bind_mutex;
std::string g_library_path;          // empty: RCCL
bool g_bound = false;

// bound once per process; never unloaded (communicators may outlive any one caller)
Rccl* rccl() {
    static Rccl* const lib = [] {
        Rccl* r = new Rccl;
        std::string path;
        { std::lock_guard<std::mutex> lock(g_bind_mutex); path = g_library_path; g_bound = true; }
        if (!path.empty()) {
            r->handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!r->handle) { r->error = path + " could not be loaded: " + (dlerror() ? dlerror() : "?"); return r; }
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                r->handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r->handle) break;
            }
        }
        if (!r->handle) { r->error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"); return r; }
        auto bind = [&](auto& fn, const char* symbol) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r->handle, symbol));
            if (!fn && r->error.empty()) r->error = std::string("librccl has no symbol ") + symbol;
        };
        bind(r->GetVersion, "ncclGetVersion"); bind(r->GetUniqueId, "ncclGetUniqueId"); bind(r->CommInitRank, "ncclCommInitRank");
        bind(r->CommDestroy, "ncclCommDestroy"); bind(r->CommAbort, "ncclCommAbort"); bind(r->AllGather, "ncclAllGather");
        bind(r->Send, "ncclSend"); bind(r->Recv, "ncclRecv"); bind(r->GroupStart, "ncclGroupStart"); bind(r->GroupEnd, "ncclGroupEnd");
        bind(r->GetErrorString, "ncclGetErrorString");
        return r;
    }();
    return lib;
}

}  // namespace

struct havac_gather {
    Rccl* lib = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 0;
    int device = 0;
    hipStream_t stream = nullptr;          // every operation of this communicator runs here, in the order it was asked for
    hipEvent_t before = nullptr, after = nullptr;
    int64_t* d_counts = nullptr;           // world + 1 words: [0..world) the gathered counts, [world] this rank's own
    int64_t* h_counts = nullptr;           // pinned, same layout
    std::vector<int64_t> counts;           // of the last havac_gather_counts
    bool counted = false, broken = false;
    bool records_pending = false;          // havac_gather_records enqueued something `after` has not been seen to pass
    uint32_t deadline_ms = 0;              // havac_gather_set_deadline: 0 = wait without limit
    std::string err;
};

#define GATHER_HIP(g, expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (g)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return _e == hipErrorOutOfMemory ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME;                 \
        }                                                                                       \
    } while (0)
#define GATHER_NCCL(g, expr)                                                                    \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            (g)->err = std::string(#expr) + ": " + (g)->lib->GetErrorString(_r);                \
            (g)->broken = true;                                                                 \
            return HAVAC_E_RUNTIME;                                                             \
        }                                                                                       \
    } while (0)

extern "C" int havac_gather_use_library(const char* path) {
    std::lock_guard<std::mutex> lock(g_bind_mutex);
    const std::string want = path ? path : "";
    if (g_bound) return want == g_library_path ? HAVAC_OK : HAVAC_E_LOGIC;      // bound already: one collective library per process
    g_library_path = want;
    return HAVAC_OK;
}

namespace {
// Waits for `event` (recorded on the communicator's stream) for at most g->deadline_ms (0: without limit).  A collective that
// does not complete -- a peer that died, or never reached it -- would otherwise hold this rank inside hipStreamSynchronize for
// ever, and with it the job's other ranks: the deadline turns that into an error that names the rank and the stage.  The
// communicator is unusable afterwards (an operation of it is still in flight: havac_gather_destroy aborts it).
int wait_with_deadline(havac_gather* g, hipEvent_t event, const char* stage) {
    if (g->deadline_ms == 0) {
        hipError_t e = hipEventSynchronize(event);
        if (e != hipSuccess) { g->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); g->broken = true; return HAVAC_E_RUNTIME; }
        return HAVAC_OK;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(g->deadline_ms);
    for (;;) {
        const hipError_t q = hipEventQuery(event);
        if (q == hipSuccess) return HAVAC_OK;
        if (q != hipErrorNotReady) { g->err = std::string("hipEventQuery: ") + hipGetErrorString(q); g->broken = true; return HAVAC_E_RUNTIME; }
        if (std::chrono::steady_clock::now() >= deadline) {
            g->err = "rank " + std::to_string(g->rank) + " of " + std::to_string(g->world) + ": " + stage + " did not complete within " +
                     std::to_string(g->deadline_ms) + " ms (a peer never reached it, or left)";
            g->broken = true;
            return HAVAC_E_TIMEOUT;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
}  // namespace

extern "C" int havac_gather_set_deadline(havac_gather* g, uint32_t timeout_ms) {
    if (!g) return HAVAC_E_ARGUMENT;
    g->deadline_ms = timeout_ms;
    return HAVAC_OK;
}

extern "C" int havac_gather_wait(havac_gather* g) {
    if (!g) return HAVAC_E_ARGUMENT;
    if (!g->records_pending) return HAVAC_OK;
    if (g->broken) { g->err = "the communicator is broken (an earlier operation failed): destroy it"; return HAVAC_E_LOGIC; }
    GATHER_HIP(g, hipSetDevice(g->device));
    const int rc = wait_with_deadline(g, g->after, "the gather of the records (ncclSend / ncclRecv)");
    if (rc == HAVAC_OK) g->records_pending = false;
    return rc;
}

extern "C" int havac_gather_rccl_version(int* version) {
    Rccl* const lib = rccl();
    if (!version) return HAVAC_E_ARGUMENT;
    if (!lib->error.empty()) return HAVAC_E_NO_DEVICE;
    return lib->GetVersion(version) == ncclSuccess ? HAVAC_OK : HAVAC_E_RUNTIME;
}

extern "C" int havac_gather_unique_id(uint8_t id[HAVAC_GATHER_ID_BYTES]) {
    static_assert(HAVAC_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES && sizeof(ncclUniqueId) == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's, byte for byte");
    Rccl* const lib = rccl();
    if (!id) return HAVAC_E_ARGUMENT;
    if (!lib->error.empty()) return HAVAC_E_NO_DEVICE;
    ncclUniqueId u;
    if (lib->GetUniqueId(&u) != ncclSuccess) return HAVAC_E_RUNTIME;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return HAVAC_OK;
}

extern "C" void havac_gather_destroy(havac_gather* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->comm) {
        // a communicator with an operation that can never complete (a peer left, a refused receive buffer) must be aborted, not drained
        // (what is still queued gets the same deadline as any other wait: a peer that died must not hold this rank in its destructor)
        if (!g->broken && g->stream && g->after && hipEventRecord(g->after, g->stream) == hipSuccess)
            (void)wait_with_deadline(g, g->after, "the communicator's last operations");
        if (g->broken) (void)g->lib->CommAbort(g->comm);
        else (void)g->lib->CommDestroy(g->comm);
    }
    if (g->before) (void)hipEventDestroy(g->before);
    if (g->after) (void)hipEventDestroy(g->after);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->d_counts) (void)hipFree(g->d_counts);
    if (g->h_counts) (void)hipHostFree(g->h_counts);
    delete g;
}

extern "C" int havac_gather_create(uint32_t rank, uint32_t world, const uint8_t id[HAVAC_GATHER_ID_BYTES], havac_gather** out) {
    if (!out || !id || world == 0 || rank >= world || world > (1u << 16)) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    Rccl* const lib = rccl();
    if (!lib->error.empty()) { std::fprintf(stderr, "havac_gather_create: %s\n", lib->error.c_str()); return HAVAC_E_NO_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HAVAC_E_NO_DEVICE;
    havac_gather* g = new (std::nothrow) havac_gather;
    if (!g) return HAVAC_E_NOMEM;
    g->lib = lib; g->rank = rank; g->world = world; g->counts.assign(world, 0);
    auto fail = [&](int code) { std::fprintf(stderr, "havac_gather_create: %s\n", g->err.c_str()); havac_gather_destroy(g); return code; };
    auto body = [&]() -> int {
        GATHER_HIP(g, hipGetDevice(&g->device));
        int least = 0, greatest = 0;
        GATHER_HIP(g, hipDeviceGetStreamPriorityRange(&least, &greatest));
        // HIGH priority, as the pipe's kernel streams (round 5; rounds 3-4: low, "the gather must not take the next kernel's compute
        // units").  The gather of pass k runs beside the SSV kernels of passes k + 1 and k + 2, which always have workgroups waiting
        // for a wave slot: a low-priority launch beside them waits 30 - 330 us on average for its first slot (measured with this
        // round's ordering kernels), and here the HOST waits for the count exchange -- eight ranks' worth of
        // such delays, the slowest one counts -- before it can submit the pass after next.  What the collective's few workgroups
        // take from an SSV kernel for the 0.2 ms of an 8 MB transfer is below a per cent; a pipeline that runs dry is not.
        GATHER_HIP(g, hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, greatest));
        GATHER_HIP(g, hipEventCreateWithFlags(&g->before, hipEventDisableTiming));
        GATHER_HIP(g, hipEventCreateWithFlags(&g->after, hipEventDisableTiming));
        GATHER_HIP(g, hipMalloc(&g->d_counts, ((size_t)world + 1) * sizeof(int64_t)));
        GATHER_HIP(g, hipHostMalloc(&g->h_counts, ((size_t)world + 1) * sizeof(int64_t), hipHostMallocDefault));
        ncclUniqueId u;
        std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        GATHER_NCCL(g, lib->CommInitRank(&g->comm, (int)world, u, (int)rank));      // collective: returns when every rank has called it
        return HAVAC_OK;
    };
    const int rc = body();
    if (rc != HAVAC_OK) return fail(rc);
    *out = g;
    return HAVAC_OK;
}

extern "C" int havac_gather_info(havac_gather* g, uint32_t* rank, uint32_t* world) {
    if (!g) return HAVAC_E_ARGUMENT;
    if (rank) *rank = g->rank;
    if (world) *world = g->world;
    return HAVAC_OK;
}

extern "C" const char* havac_gather_last_error(havac_gather* g) { return g ? g->err.c_str() : "null gather handle"; }

// Step 1: every rank tells how many records its pass left (a negative count: the pass failed).  The one host wait of a gather.
extern "C" int havac_gather_counts(havac_gather* g, int64_t my_count, int64_t* counts_out, void* hip_stream) {
    if (!g || !counts_out) return HAVAC_E_ARGUMENT;
    if (g->broken) { g->err = "the communicator is broken (an earlier operation failed): destroy it"; return HAVAC_E_LOGIC; }
    const hipStream_t caller = (hipStream_t)hip_stream;
    GATHER_HIP(g, hipSetDevice(g->device));
    g->counted = false;
    // behind whatever the caller has queued (its pass's ordering), on the communicator's own stream
    GATHER_HIP(g, hipEventRecord(g->before, caller));
    GATHER_HIP(g, hipStreamWaitEvent(g->stream, g->before, 0));
    g->h_counts[g->world] = my_count;
    GATHER_HIP(g, hipMemcpyAsync(g->d_counts + g->world, g->h_counts + g->world, sizeof(int64_t), hipMemcpyHostToDevice, g->stream));
    GATHER_NCCL(g, g->lib->AllGather(g->d_counts + g->world, g->d_counts, 1, ncclInt64, g->comm, g->stream));
    GATHER_HIP(g, hipMemcpyAsync(g->h_counts, g->d_counts, (size_t)g->world * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
    GATHER_HIP(g, hipEventRecord(g->after, g->stream));
    if (const int rc = wait_with_deadline(g, g->after, "the exchange of the counts (ncclAllGather)")) return rc;
    g->records_pending = false;            // (the stream is in order: whatever was queued before the counts has passed too)
    for (uint32_t r = 0; r < g->world; r++) counts_out[r] = g->counts[r] = g->h_counts[r];
    g->counted = true;
    return HAVAC_OK;
}

// Step 2: the records.  Rank r > 0 sends its count[r] records; rank 0 receives every list at its offset of d_out and copies
// its own.  Enqueued; `hip_stream` is made to wait for it, the host is not.
extern "C" int havac_gather_records(havac_gather* g, const uint64_t* d_records, uint64_t* d_out, uint64_t out_capacity, void* hip_stream) {
    if (!g) return HAVAC_E_ARGUMENT;
    if (g->broken) { g->err = "the communicator is broken (an earlier operation failed): destroy it"; return HAVAC_E_LOGIC; }
    if (!g->counted) { g->err = "havac_gather_records needs the counts of this pass: call havac_gather_counts first"; return HAVAC_E_LOGIC; }
    g->counted = false;
    uint64_t total = 0;
    for (uint32_t r = 0; r < g->world; r++) {
        if (g->counts[r] < 0) { g->err = "the pass failed on rank " + std::to_string(r) + ": nothing to gather"; return HAVAC_E_RUNTIME; }   // the same on every rank
        total += (uint64_t)g->counts[r];
    }
    const uint64_t mine = (uint64_t)g->counts[g->rank];
    if (mine && !d_records) { g->err = "no records given"; g->broken = true; return HAVAC_E_ARGUMENT; }
    if (g->rank == 0 && total && (!d_out || out_capacity < total)) {
        // (refused before anything is posted; the other ranks' sends can never complete now: the communicator is done for)
        g->err = "receive buffer of " + std::to_string(out_capacity) + " records for " + std::to_string(total);
        g->broken = true;
        return HAVAC_E_LENGTH;
    }
    const hipStream_t caller = (hipStream_t)hip_stream;
    GATHER_HIP(g, hipSetDevice(g->device));
    GATHER_HIP(g, hipEventRecord(g->before, caller));
    GATHER_HIP(g, hipStreamWaitEvent(g->stream, g->before, 0));
    if (g->world > 1) {
        GATHER_NCCL(g, g->lib->GroupStart());
        // (a failure between GroupStart and GroupEnd must not leave the thread's group open: the group is closed first, the
        // error reported afterwards -- the communicator is broken either way, and havac_gather_destroy aborts it)
        ncclResult_t posted = ncclSuccess;
        const char* what = "";
        if (g->rank != 0) {
            if (mine) { posted = g->lib->Send(d_records, (size_t)mine, ncclUint64, 0, g->comm, g->stream); what = "ncclSend"; }
        } else {
            uint64_t offset = mine;
            for (uint32_t r = 1; r < g->world && posted == ncclSuccess; r++) {
                const uint64_t n = (uint64_t)g->counts[r];
                if (n) { posted = g->lib->Recv(d_out + offset, (size_t)n, ncclUint64, (int)r, g->comm, g->stream); what = "ncclRecv"; }
                offset += n;
            }
        }
        const ncclResult_t closed = g->lib->GroupEnd();
        if (posted != ncclSuccess || closed != ncclSuccess) {
            const bool in_group = posted != ncclSuccess;
            g->err = std::string(in_group ? what : "ncclGroupEnd") + ": " + g->lib->GetErrorString(in_group ? posted : closed);
            g->broken = true;
            return HAVAC_E_RUNTIME;
        }
    }
    if (g->rank == 0 && mine && d_out != d_records)
        GATHER_HIP(g, hipMemcpyAsync(d_out, d_records, (size_t)mine * sizeof(uint64_t), hipMemcpyDeviceToDevice, g->stream));
    GATHER_HIP(g, hipEventRecord(g->after, g->stream));
    GATHER_HIP(g, hipStreamWaitEvent(caller, g->after, 0));
    g->records_pending = true;
    return HAVAC_OK;
}
