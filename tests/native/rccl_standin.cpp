// rccl_standin.cpp -- TEST INFRASTRUCTURE: the eleven RCCL entry points libhavac_dev.so binds (havac_amd/csrc/havac_gather.hip:
// ncclGetVersion, ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommAbort, ncclAllGather, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclGetErrorString) for several PROCESSES that share ONE GPU.
//
// Why: RCCL refuses two ranks on one device, and this pool hands out one-GPU boxes, so the N > 1 branch of the C-ABI gather
// (the grouped ncclSend / ncclRecv of havac_gather_records) could never run before the driver's 8-GPU job did.  With this
// library handed to havac_gather_use_library() the real havac_gather_* code -- offsets, zero-length ranks, the -1 sentinel, the
// refused receive buffer, the deadline -- runs with 2 ... 8 processes on one card.  Nothing of the product links or loads it.
//
// How: a POSIX shared-memory segment named by the unique id holds one single-slot mailbox per ordered pair of ranks; a message
// crosses in chunks (device -> mailbox by the sender, mailbox -> device by the receiver, plain hipMemcpy).  Operations complete
// inside the call (after the stream they were "enqueued" on has been waited for), which is stronger than RCCL's stream
// semantics and therefore valid for every caller of those; ncclGroupStart / ncclGroupEnd collect the operations and run them
// together, so a group that sends and receives cannot deadlock on itself.  A wait that sees no progress for
// `standin_set_timeout_ms` (default 60 s) fails with ncclSystemError: a test never hangs.  `standin_set_delay_ms` makes the
// next collective leave a BOUNDED spinning kernel on the stream behind its work (it ends by itself after that time): what an
// operation still in flight looks like to the caller's deadline.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

constexpr uint32_t kMaxRanks = 8;
constexpr size_t kChunk = 256 * 1024;

struct Mailbox {                      // rank `src` -> rank `dst`, one chunk at a time
    std::atomic<uint64_t> produced, consumed;
    uint32_t len;
    alignas(64) uint8_t data[kChunk];
};
struct Segment {
    std::atomic<uint32_t> ready, attached, left;
    uint32_t world;
    Mailbox box[kMaxRanks][kMaxRanks];
};

std::atomic<uint32_t> g_timeout_ms{60000}, g_delay_ms{0};

struct Op { bool send; int peer; uint8_t* ptr; size_t bytes, done; hipStream_t stream; };
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local struct ncclComm* t_group_comm = nullptr;

// a kernel that keeps the stream busy for `ms` milliseconds and then ends (wall_clock64: 100 MHz): bounded by construction
__global__ void standin_delay(unsigned long long ticks) {
    const unsigned long long start = wall_clock64();
    while (wall_clock64() - start < ticks) __builtin_amdgcn_s_sleep(64);
}

}  // namespace

struct ncclComm {
    Segment* seg = nullptr;
    int rank = 0, world = 0;
    std::string name;
};

namespace {

void maybe_delay(hipStream_t stream);

bool run_ops(ncclComm* c, std::vector<Op>& ops) {
    // behind what the streams hold (the caller's records are ordered there)
    for (const Op& op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return false;
    auto last_progress = std::chrono::steady_clock::now();
    for (;;) {
        bool all_done = true, moved = false;
        bool channel_busy_send[kMaxRanks] = {}, channel_busy_recv[kMaxRanks] = {};
        for (Op& op : ops) {
            if (op.done == op.bytes) continue;
            all_done = false;
            // messages of one channel go in the order they were posted
            bool& busy = op.send ? channel_busy_send[op.peer] : channel_busy_recv[op.peer];
            if (busy) continue;
            busy = true;
            Mailbox& m = op.send ? c->seg->box[c->rank][op.peer] : c->seg->box[op.peer][c->rank];
            if (op.send) {
                if (m.consumed.load(std::memory_order_acquire) != m.produced.load(std::memory_order_relaxed)) continue;
                const size_t n = std::min(kChunk, op.bytes - op.done);
                if (hipMemcpy(m.data, op.ptr + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) return false;
                m.len = (uint32_t)n;
                m.produced.fetch_add(1, std::memory_order_release);
                op.done += n; moved = true;
            } else {
                if (m.produced.load(std::memory_order_acquire) == m.consumed.load(std::memory_order_relaxed)) continue;
                const size_t n = m.len;
                if (n > op.bytes - op.done) return false;                       // the sender posted more than this receive takes
                if (hipMemcpy(op.ptr + op.done, m.data, n, hipMemcpyHostToDevice) != hipSuccess) return false;
                m.consumed.fetch_add(1, std::memory_order_release);
                op.done += n; moved = true;
            }
        }
        if (all_done) return true;
        const auto now = std::chrono::steady_clock::now();
        if (moved) last_progress = now;
        else if (now - last_progress > std::chrono::milliseconds(g_timeout_ms.load())) return false;
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

ncclResult_t post(ncclComm* c, Op op) {
    if (!c || !c->seg || op.peer < 0 || op.peer >= c->world || op.peer == c->rank) return ncclInvalidArgument;
    if (t_group_depth > 0) {
        if (t_group_comm && t_group_comm != c) return ncclInvalidArgument;
        t_group_comm = c;
        t_ops.push_back(op);
        return ncclSuccess;
    }
    std::vector<Op> one{op};
    if (!run_ops(c, one)) return ncclSystemError;
    maybe_delay(op.stream);
    return ncclSuccess;
}

void maybe_delay(hipStream_t stream) {
    const uint32_t ms = g_delay_ms.exchange(0);
    if (ms) hipLaunchKernelGGL(standin_delay, dim3(1), dim3(64), 0, stream, (unsigned long long)ms * 100000ull);
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}  // namespace

extern "C" {

// test knobs (not part of RCCL)
void standin_set_timeout_ms(uint32_t ms) { g_timeout_ms.store(ms); }
void standin_set_delay_ms(uint32_t ms) { g_delay_ms.store(ms); }

ncclResult_t ncclGetVersion(int* version) {
    if (!version) return ncclInvalidArgument;
    *version = 22203;                     // "2.22.3": what a caller prints; says nothing about this library
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    std::random_device rd;
    std::snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/havac_rccl_standin_%d_%08x%08x", (int)getpid(), (unsigned)rd(), (unsigned)rd());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > (int)kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (std::strncmp(id.internal, "/havac_rccl_standin_", 20) != 0) return ncclInvalidArgument;
    ncclComm* c = new ncclComm;
    c->rank = rank; c->world = nranks; c->name.assign(id.internal, strnlen(id.internal, NCCL_UNIQUE_ID_BYTES));
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(g_timeout_ms.load());
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    } else {
        for (;;) {
            fd = shm_open(c->name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == sizeof(Segment)) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::steady_clock::now() > deadline) { delete c; return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (rank == 0) shm_unlink(c->name.c_str()); delete c; return ncclSystemError; }
    c->seg = static_cast<Segment*>(p);
    if (rank == 0) {                      // (a fresh segment is all zeroes: every counter starts at 0)
        c->seg->world = (uint32_t)nranks;
        c->seg->ready.store(1, std::memory_order_release);
    } else {
        while (c->seg->ready.load(std::memory_order_acquire) == 0) {
            if (std::chrono::steady_clock::now() > deadline) { munmap(p, sizeof(Segment)); delete c; return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        if (c->seg->world != (uint32_t)nranks) { munmap(p, sizeof(Segment)); delete c; return ncclInvalidArgument; }
    }
    c->seg->attached.fetch_add(1, std::memory_order_acq_rel);
    // returns when every rank has called it, as ncclCommInitRank does; then the name can go: the mappings keep the memory
    while (c->seg->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (std::chrono::steady_clock::now() > deadline) { if (rank == 0) shm_unlink(c->name.c_str()); munmap(p, sizeof(Segment)); delete c; return ncclSystemError; }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (rank == 0) shm_unlink(c->name.c_str());
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    if (c->seg) { c->seg->left.fetch_add(1, std::memory_order_acq_rel); munmap(c->seg, sizeof(Segment)); }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclSystemError: return "stand-in RCCL: an operation made no progress before its deadline (a peer never posted its side), or a copy failed";
        case ncclInvalidArgument: return "stand-in RCCL: invalid argument";
        default: return "stand-in RCCL: error";
    }
}

ncclResult_t ncclGroupStart() { t_group_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidArgument;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    ncclComm* const c = t_group_comm;
    t_group_comm = nullptr;
    if (ops.empty()) return ncclSuccess;
    if (!run_ops(c, ops)) return ncclSystemError;
    maybe_delay(ops[0].stream);
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    return post(c, Op{true, peer, const_cast<uint8_t*>(static_cast<const uint8_t*>(buf)), count * type_bytes(type), 0, stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    return post(c, Op{false, peer, static_cast<uint8_t*>(buf), count * type_bytes(type), 0, stream});
}

// every rank's block to every rank: world - 1 sends and world - 1 receives run together, the own block copied in place
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
    if (!c || !c->seg) return ncclInvalidArgument;
    const size_t bytes = count * type_bytes(type);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclSystemError;
    uint8_t* const out = static_cast<uint8_t*>(recvbuff);
    if (out + (size_t)c->rank * bytes != sendbuff &&
        hipMemcpy(out + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclSystemError;
    std::vector<Op> ops;
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        ops.push_back(Op{true, r, const_cast<uint8_t*>(static_cast<const uint8_t*>(sendbuff)), bytes, 0, stream});
        ops.push_back(Op{false, r, out + (size_t)r * bytes, bytes, 0, stream});
    }
    if (!run_ops(c, ops)) return ncclSystemError;
    maybe_delay(stream);
    return ncclSuccess;
}

}  // extern "C"
