// hit_order.hip.h -- the hit queue put into the FPGA's emission order without a generic sort, in ONE launch.
//
// The reference's device emits its records segment by segment, row by row, column by column
// (device/HavacHls.cpp:151-152,264; the sieve of device/HitReporting.cpp:178-337 keeps the columns ascending).  The SSV
// kernel here appends records in the order its 6,144 waves find them, as sort keys
//     key = segment << (14 + row_bits) | row << 14 | column-in-segment        (ssv_kernels.hip.h: hit_key)
// so device order = ascending key, and keys are unique (a cell hits at most once).  Round 2 ordered them with rocPRIM's
// radix sort: 5-6 Onesweep passes over all keys whatever their structure.  Round 3 used the structure, in five kernels:
//
//   * a BUCKET is a run of consecutive key values: (key >> shift) -- a group of 2^g segments where segments hold few records,
//     one segment, or one range of 2^k rows of a segment when segments hold many (tall collections).  Buckets in ascending
//     order concatenate to the sorted whole, so only the inside of a bucket has to be sorted;
//   * count: one pass over the keys, one non-returning atomic per RUN of equal buckets inside a wave (a burst of the SSV
//     kernel's queue comes from one tile: runs are long);
//   * scan: exclusive scan of the counts, the list of buckets too big for the small sorter, and the `oversized` flag when a
//     bucket exceeds what an LDS sort holds;
//   * scatter: second pass over the keys, each run to its bucket's stretch of a second buffer (one returning atomic per run);
//   * sort: one WAVE per bucket of up to 256 records sorts the keys' low bits (32-bit) in registers -- a bitonic network whose
//     short-distance stages are cross-lane reads and whose long-distance stages are register exchanges, no LDS and no
//     barrier -- and writes the bucket back as the reference's packed RECORDS (device/HitReporting.cpp:421-430) at its final
//     place; the rare buckets of 257 ... 16,384 records: one workgroup each, bitonic in LDS.
//
// Round 5: nothing of it waits for the host any more.  A pass of the reference's one-run-at-a-time API (host/HavacHwClient.cpp:
// 141-157,172-202: run, wait, list) paid for eight small launches, two copies and a host round trip in the middle -- the count
// had to reach the host before the ordering could be sized: 72 us behind a 130 us kernel on a 64-row model.
// Now the kernels read the count on the device -- behind ssv_gather_tails, which still appends the blocks' tails to the queue
// (walking the tails where they lie, twice, cost more than compacting them once: profiles/r05e_*) -- pick the bucket width
// themselves (order_shape), and the last kernel's last workgroup writes the outcome into pinned host memory.  Five launches (count | scan | scatter | sort | the large buckets and the report), enqueued with the SSV kernel.  What the kernels cannot do with what they were given -- a second buffer, bucket tables or an
// LDS sorter too small for this pass -- is reported without anything having been moved, and the host grows the buffer and
// launches them again (havac_dev.hip).  A bucket above kLargeBucket records (a clump far denser than the launch's average,
// e.g. one low-complexity model in a sparse collection) sends the pass to the generic radix sort, as before.
// Tried first and dropped (profiles/r05c_order_probe.txt): ONE cooperative launch with two grid-wide barriers -- a barrier of
// 256 workgroups (release, one atomic each, a bounded spin, acquire) costs 9 us, more than twice a kernel boundary, and grows
// with the grid: 76 us for a list that the separate kernels order in 40.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "ssv_kernels.hip.h"

namespace havac {

constexpr uint32_t kSmallBucket = 256;       // records one WAVE sorts in registers (4 per lane)
constexpr uint32_t kLargeBucket = 16384;     // records the large sorter can hold (64 KB of LDS, one workgroup): the launch with the big LDS
constexpr uint32_t kLargeBucketDefault = 2048;   // ... and the usual launch (8 KB: the kernel fits beside six workgroups of either SSV kernel)
constexpr uint32_t kTargetBucket = 128;      // average records per bucket the kernel aims for when it picks the shift
constexpr uint32_t kScanChunk = 2048;        // buckets per step of the scan (256 threads x 8)
constexpr uint32_t kOrderThreads = 256;

// device words the workgroups of one ssv_order_pass share (cleared by ssv_prepare_model in front of the pass's SSV kernel)
struct OrderState {
    uint32_t counted;                   // workgroups of the scan that have written their chunk's total: the last one scans the totals
    uint32_t done;                      // workgroups of the last kernel that have left: the last one reports to the host
    uint32_t pad0[4];
    uint32_t oversized;                 // 1: a bucket exceeds kLargeBucket; 2: the counts do not add up to the pass's records
    uint32_t nlarge, largest;           // buckets above kSmallBucket (listed in large_list); the largest bucket
    uint32_t pad[7];
};
static_assert(sizeof(OrderState) == 64, "one cache line");

// What a pass's ordering tells the host (pinned memory, written by the kernel's last workgroup).
enum : uint32_t {
    kOrderNone = 0,           // (the host's value before a launch)
    kOrderDone = 1,           // the records are in d_hits[0, found) in device order
    kOrderOverflow = 2,       // more records than the hit buffer holds: nothing ordered
    kOrderRetry = 3,          // a buffer of the context is too small for this pass (need_*): nothing moved, launch again
    kOrderGeneric = 4         // a bucket too big for any LDS sort, counts that do not add up, or the caller asked for the
                              // generic sort: nothing moved -- gather the tails, radix sort
};
struct PassReport {
    unsigned long long found;       // records the pass found: queue + tails (may exceed the capacity)
    unsigned long long need_alt;    // kOrderRetry: records the second buffer must hold
    uint32_t status, need_buckets, need_large;
    uint32_t nbuckets, largest, nlarge, oversized, shift;
    uint32_t ssv_fault;             // the SSV kernel's fault word (a row-block hand-off that never came)
    uint32_t pad;
};

struct OrderPass {                  // the kernel's argument
    uint64_t* hits;                 // in: the queue's sort keys; out: the records in device order
    uint64_t* alt; unsigned long long alt_capacity;
    unsigned long long capacity;    // of `hits`
    const unsigned long long* hit_count;     // records the pass appended to the queue, the blocks' tails included (may run past the capacity)
    uint32_t* counts; uint32_t* offsets; uint64_t* chunk_base; uint32_t* large_list; uint32_t max_buckets;
    OrderState* state; PassReport* host; const uint32_t* ssv_fault;
    uint32_t row_bits, seg_bits, large_capacity /* keys the launch's LDS holds */, generic /* 1: report kOrderGeneric, move nothing */;
    unsigned long long first_segment, nsegments;
};

// runs of equal bucket numbers among the active lanes of a wave: a run is a stretch of consecutive ACTIVE lanes with the same
// bucket (an inactive lane -- one beyond the list's end -- ends it; it carries bucket ~0)
struct WaveRuns { unsigned long long heads; uint32_t head_lane, run_length; bool is_head; };
__device__ __forceinline__ WaveRuns wave_runs(uint32_t bucket, bool active) {
    const uint32_t lane = __lane_id();
    const uint32_t before = __shfl_up(bucket, 1, 64);
    WaveRuns r;
    r.is_head = active && (lane == 0 || before != bucket);
    r.heads = __ballot(r.is_head);
    const unsigned long long enders = r.heads | ~__ballot(active);                         // where a run may end: the next head, or a hole
    const unsigned long long upto_me = r.heads & ((2ull << lane) - 1ull);                // heads at or before this lane
    r.head_lane = upto_me ? 63u - (uint32_t)__clzll(upto_me) : 0u;
    const unsigned long long after_head = enders & ~((2ull << r.head_lane) - 1ull);
    const uint32_t run_end = after_head ? (uint32_t)__builtin_ctzll(after_head) : 64u;
    r.run_length = run_end - r.head_lane;
    return r;
}

constexpr int kKeysPerLane = 4;              // count: keys a lane has in flight
constexpr int kScatterKeysPerLane = 2;       // scatter (more would cost the 32-VGPR budget)

__device__ __forceinline__ uint64_t bucket_begin(const uint64_t* __restrict__ chunk_base, const uint32_t* __restrict__ local_offsets, uint32_t b) {
    return chunk_base[b / kScanChunk] + local_offsets[b];
}

// ---- small buckets: one WAVE sorts one bucket in registers ----------------------------------------------------------------
// Bitonic network over N = 64 K values, element e = 64 r + lane in register r of lane `lane`: a stage with partner distance
// j < 64 exchanges between lanes, a stage with j >= 64 between registers of the same lane.  No LDS memory, no barrier; a
// workgroup's four waves sort four buckets independently.
// The network is unrolled completely, so everything about a stage is known at compile time: the partner (a DPP quad
// permutation for distances 1 and 2, ds_swizzle for 4, 8, 16 -- no address arithmetic), and WHICH lanes keep the smaller value
// -- a 64-bit constant handed to v_cndmask in an SGPR pair.  A compare-exchange is then one cross-lane move, v_min, v_max
// and v_cndmask.  That matters because this kernel runs beside the SSV kernel of the following pass, which is bound by VALU
// issue: written with run-time distances (address arithmetic for ds_bpermute, lane predicates recomputed per stage: a dozen
// instructions per exchange) the ordering of a C2 pass cost that kernel 20 us; see DESIGN.md section 4.3.
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);              // bit-mask mode: and 0x1f, xor 4
    else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);
    else if constexpr (J == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    else return (uint32_t)__shfl_xor((int)v, 32, 64);
}
// lanes of register R that keep the SMALLER value in stage (k, j), j < 64: the lower partner of an ascending pair, the upper of a descending one
template <uint32_t R, uint32_t Kk, uint32_t J>
constexpr uint64_t keep_min_lanes() {
    uint64_t m = 0;
    for (uint32_t lane = 0; lane < 64; lane++) {
        const bool lower = (lane & J) == 0, up = ((R * 64u + lane) & Kk) == 0;
        if (lower == up) m |= 1ull << lane;
    }
    return m;
}
template <int K, uint32_t Kk, uint32_t J, int R>
__device__ __forceinline__ void lane_exchange(uint32_t (&x)[K]) {
    constexpr uint64_t keep_min = keep_min_lanes<(uint32_t)R, Kk, J>();
    const uint32_t mine = x[R], other = lane_xor<J>(mine);
    const uint32_t lo = mine < other ? mine : other, hi = mine < other ? other : mine;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(x[R]) : "v"(hi), "v"(lo), "s"(keep_min));
}
template <int K, uint32_t Kk, uint32_t J, int... R>
__device__ __forceinline__ void lane_stage(uint32_t (&x)[K], std::integer_sequence<int, R...>) {
    (lane_exchange<K, Kk, J, R>(x), ...);
}
template <int K, uint32_t Kk, uint32_t J>
__device__ __forceinline__ void bitonic_stage(uint32_t (&x)[K]) {
    if constexpr (J >= 64) {
        constexpr int JR = J / 64;
#pragma unroll
        for (int r = 0; r < K; r++) {
            if ((r & JR) == 0) {                                   // the direction of a pair of registers does not depend on the lane
                const bool up = (((uint32_t)r * 64u) & Kk) == 0;
                const uint32_t a = x[r], b = x[r | JR];
                const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                x[r] = up ? lo : hi;
                x[r | JR] = up ? hi : lo;
            }
        }
    } else {
        lane_stage<K, Kk, J>(x, std::make_integer_sequence<int, K>{});
    }
    if constexpr (J > 1) bitonic_stage<K, Kk, J / 2>(x);
}
template <int K, uint32_t Kk = 2>
__device__ __forceinline__ void wave_bitonic_sort(uint32_t (&x)[K]) {
    bitonic_stage<K, Kk, Kk / 2>(x);
    if constexpr (Kk < 64u * K) wave_bitonic_sort<K, Kk * 2>(x);
}
template <int K>
__device__ __forceinline__ void sort_bucket_in_a_wave(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t begin, uint32_t count,
                                                      uint64_t bucket_key, uint64_t low_mask, uint32_t row_bits, uint32_t lane) {
    uint32_t x[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        const uint32_t e = (uint32_t)r * 64u + lane;
        x[r] = e < count ? (uint32_t)(in[begin + e] & low_mask) : 0xffffffffu;
    }
    wave_bitonic_sort<K>(x);
#pragma unroll
    for (int r = 0; r < K; r++) {
        const uint32_t e = (uint32_t)r * 64u + lane;
        if (e < count) out[begin + e] = key_to_record(bucket_key | x[r], row_bits);
    }
}

template <uint32_t T>
__device__ __forceinline__ void bitonic_sort_lds(uint32_t* v, uint32_t N) {
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += T) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                const uint32_t a = v[i], b = v[i | j];
                const bool ascending = (i & k) == 0;
                if ((a > b) == ascending) { v[i] = b; v[i | j] = a; }
            }
            __syncthreads();
        }
    }
}


// a wave-uniform value, moved to scalar registers (a value loaded through a pointer arrives in vector registers even when every
// lane loaded the same word; the kernel below is held to 32 VGPRs)
__device__ __forceinline__ uint32_t uniform32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ unsigned long long uniform64(unsigned long long x) {
    return ((unsigned long long)uniform32((uint32_t)(x >> 32)) << 32) | uniform32((uint32_t)x);
}

typedef const __attribute__((address_space(4))) OrderPass* order_args_t;
__device__ __forceinline__ order_args_t order_args() {
    order_args_t p = (order_args_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// The scan between count and scatter.  Workgroup c takes buckets [c * kScanChunk, (c+1) * kScanChunk): it writes each bucket's
// offset INSIDE the chunk (offsets), lists the chunk's large buckets, clears the counts (the scatter uses them as cursors and
// leaves the counts in them again) and notes the chunk's total; the workgroup that finishes LAST scans the totals into
// chunk_base (64-bit: a pass may hold more than 2^32 records).  A bucket begins at chunk_base[b / kScanChunk] + offsets[b].
// One launch whatever the number of buckets: the grid covers the context's bucket tables, and a workgroup beyond this pass's
// buckets leaves at once.
__device__ __forceinline__ void scan_bucket_counts(uint32_t* lds, uint32_t nbuckets, unsigned long long found, uint32_t wave) {
    const order_args_t args = order_args();
    uint32_t* __restrict__ const counts = args->counts;
    uint32_t* __restrict__ const offsets = args->offsets;
    uint64_t* const chunk_base = args->chunk_base;
    uint32_t* __restrict__ const large_list = args->large_list;
    OrderState* const st = args->state;
    const uint32_t nchunks = (nbuckets + kScanChunk - 1) / kScanChunk;
    if (blockIdx.x >= nchunks) return;
    const uint32_t lane = fresh_lane(), t = lane + 64u * wave;
    uint32_t* const tile = lds;                       // kScanChunk words
    uint32_t* const wave_sum = lds + kScanChunk;      // 4 words
    __shared__ uint32_t last_block;
    const uint32_t first = blockIdx.x * kScanChunk;
    uint32_t most = 0;
    {
        uint32_t n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {                   // coalesced: thread t takes buckets first + t + 256 i; eight loads in flight
            const uint32_t b = first + t + 256u * i;
            n[i] = b < nbuckets ? counts[b] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t b = first + t + 256u * i;
            tile[t + 256u * i] = n[i];
            most = n[i] > most ? n[i] : most;
            if (n[i]) {
                counts[b] = 0;
                if (n[i] > kSmallBucket && n[i] <= kLargeBucket) large_list[atomicAdd(&st->nlarge, 1u)] = b;
            }
        }
    }
    __syncthreads();
    uint32_t sum = 0;                                   // thread t scans buckets first + 8 t ... 8 t + 7
#pragma unroll
    for (int i = 0; i < 8; i++) sum += tile[8u * t + i];
    uint32_t inclusive = sum;
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
        const uint32_t up = __shfl_up(inclusive, d, 64);
        if ((int)lane >= d) inclusive += up;
    }
    if (lane == 63) wave_sum[wave] = inclusive;
    __syncthreads();
    uint32_t before = inclusive - sum;
    for (uint32_t w = 0; w < wave; w++) before += wave_sum[w];
    const uint32_t chunk_total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint32_t mine = tile[8u * t + i]; tile[8u * t + i] = before; before += mine; }      // (each thread its own eight words)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t b = first + t + 256u * i;
        if (b < nbuckets) offsets[b] = tile[t + 256u * i];
    }
    // the largest bucket of the chunk
#pragma unroll
    for (int d = 32; d >= 1; d /= 2) { const uint32_t other = __shfl_xor(most, d, 64); most = other > most ? other : most; }
    if (lane == 0 && most) {
        atomicMax(&st->largest, most);
        if (most > kLargeBucket) atomicOr(&st->oversized, 1u);
    }
    // this chunk's total; the last workgroup to get here scans the totals
    if (t == 255) {
        chunk_base[blockIdx.x] = (uint64_t)chunk_total;          // a total for now, a base after the scan below
        __threadfence();
        last_block = __hip_atomic_fetch_add(&st->counted, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nchunks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_block) return;
    __threadfence();
    uint64_t* const part = reinterpret_cast<uint64_t*>(lds);      // 256 words of 64 bits (the tile is done with)
    const uint32_t per = (nchunks + 255u) / 256u;
    uint64_t total = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t c = t * per + k; if (c < nchunks) total += __hip_atomic_load(&chunk_base[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    part[t] = total;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d *= 2) {
        const uint64_t add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint64_t at = part[t] - total;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t c = t * per + k;
        if (c < nchunks) { const uint64_t mine_total = __hip_atomic_load(&chunk_base[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); chunk_base[c] = at; at += mine_total; }
    }
    if (t == 255) {
        chunk_base[nchunks] = part[255];
        // The counts must add up to the records of this pass.  They do unless a pass before this one was cut short (a HIP error
        // between its kernels) and left counts behind: then nothing is moved -- offsets from such counts would point outside the
        // buffers -- and the host clears the counts and orders this pass with the radix sort (`oversized` = 2).
        if (part[255] != found) atomicOr(&st->oversized, 2u);
    }
}

// ---- the pass's ordering, one launch ----------------------------------------------------------------------------------------
// Dynamic LDS: max(kScanChunk + 64 words for the scan, large_capacity words for the large sorter).
//
// Every phase is a function that reads the arguments IT needs from the kernarg segment (scalar loads through a pointer made opaque,
// as ssv_kernels.hip.h does with SsvRare) and works out the lane number afresh (fresh_lane): nothing but a handful of scalars
// (OrderShape) lives from one phase to the next, whether the phases are one kernel or four (see the kernels at the end).
struct OrderShape {                 // wave-uniform, worked out by every workgroup for itself from the same words
    unsigned long long found, n_queue, base;
    uint32_t shift, nbuckets, status;
};

__device__ __forceinline__ OrderShape order_shape() {
    const order_args_t a = order_args();
    const unsigned long long capacity = a->capacity, alt_capacity = a->alt_capacity;
    OrderShape s;
    s.found = uniform64(*a->hit_count);
    s.n_queue = s.found < capacity ? s.found : capacity;
    // bucket = key >> shift.  The widest bucket still sorts by the low 32 bits of its keys; the narrowest is one row of one
    // segment (at most 12288 records).  From the widest down until there are about found / kTargetBucket buckets.
    const uint32_t row_bits = a->row_bits;
    const uint32_t key_bits = 14u + row_bits + a->seg_bits;
    const unsigned long long first_segment = a->first_segment, nsegments = a->nsegments;
    const unsigned long long first_key = first_segment << (14u + row_bits);
    const unsigned long long last_key = ((first_segment + nsegments) << (14u + row_bits)) - 1ull;
    const unsigned long long want = s.found / kTargetBucket;
    uint32_t shift = key_bits < 32u ? key_bits : 32u;
    while (shift > 14u && (last_key >> shift) - (first_key >> shift) + 1ull < want) shift--;
    s.shift = shift;
    s.base = first_key >> shift;
    const unsigned long long nbuckets64 = (last_key >> shift) - s.base + 1ull;
    s.status = kOrderNone;
    if (s.found == 0) s.status = kOrderDone;
    else if (s.found > capacity) s.status = kOrderOverflow;
    else if (a->generic || nbuckets64 >= (1ull << 27)) s.status = kOrderGeneric;
    else if (nbuckets64 > a->max_buckets || s.found > alt_capacity) s.status = kOrderRetry;
    s.nbuckets = nbuckets64 < (1ull << 27) ? (uint32_t)nbuckets64 : 0u;
    return s;
}

// the last workgroup to leave tells the host (system-scope stores into pinned memory; the host looks after the stream's event)
__device__ __forceinline__ void order_leave(const OrderShape& s, uint32_t how, uint32_t wave) {
    __shared__ uint32_t s_last_out;
    const order_args_t a = order_args();
    OrderState* const st = a->state;
    const bool first_thread = wave == 0 && fresh_lane() == 0;
    __syncthreads();
    if (first_thread) s_last_out = __hip_atomic_fetch_add(&st->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last_out || !first_thread) return;
    PassReport* const h = a->host;
    h->found = s.found;
    h->need_alt = s.found;
    h->need_buckets = s.nbuckets;
    h->nbuckets = s.nbuckets;
    h->largest = __hip_atomic_load(&st->largest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->need_large = h->largest;
    h->nlarge = __hip_atomic_load(&st->nlarge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->oversized = __hip_atomic_load(&st->oversized, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->shift = s.shift;
    const uint32_t* const ssv_fault = a->ssv_fault;
    h->ssv_fault = ssv_fault ? *ssv_fault : 0u;
    __hip_atomic_store(&h->status, how, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// count: one non-returning atomic per run of equal buckets inside a wave
__device__ __forceinline__ void order_count(const OrderShape& s, uint64_t gwave, uint64_t nwaves) {
    const order_args_t a = order_args();
    const uint64_t* __restrict__ const keys = a->hits;
    const uint64_t n = s.n_queue;
    uint32_t* const counts = a->counts;
    const uint32_t lane = fresh_lane();
    const uint64_t span = 64u * kKeysPerLane;
    for (uint64_t first = gwave * span; first < n; first += nwaves * span) {
        uint64_t key[kKeysPerLane];
#pragma unroll
        for (int u = 0; u < kKeysPerLane; u++) { const uint64_t i = first + 64u * u + lane; key[u] = i < n ? keys[i] : 0; }
#pragma unroll
        for (int u = 0; u < kKeysPerLane; u++) {
            const bool active = first + 64u * u + lane < n;
            const uint32_t bucket = active ? (uint32_t)((key[u] >> s.shift) - s.base) : 0xffffffffu;
            const WaveRuns r = wave_runs(bucket, active);
            if (r.is_head) atomicAdd(&counts[bucket], r.run_length);
        }
    }
}

// scatter: each run to its bucket's stretch of the second buffer (one returning atomic per run)
__device__ __forceinline__ void order_scatter(const OrderShape& s, uint64_t gwave, uint64_t nwaves) {
    const order_args_t a = order_args();
    const uint64_t* __restrict__ const keys = a->hits;
    const uint64_t n = s.n_queue;
    uint64_t* __restrict__ const alt = a->alt;
    uint32_t* const counts = a->counts;
    const uint64_t* __restrict__ const chunk_base = a->chunk_base;
    const uint32_t* __restrict__ const offsets = a->offsets;
    const uint32_t lane = fresh_lane();
    const uint64_t span = 64u * kScatterKeysPerLane;
    for (uint64_t first = gwave * span; first < n; first += nwaves * span) {
        uint64_t key[kScatterKeysPerLane];
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) { const uint64_t i = first + 64u * u + lane; key[u] = i < n ? keys[i] : 0; }
        uint64_t at[kScatterKeysPerLane];
        WaveRuns runs[kScatterKeysPerLane];
        bool active[kScatterKeysPerLane];
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) {          // the returning atomics of a lane's heads are in flight together
            active[u] = first + 64u * u + lane < n;
            const uint32_t bucket = active[u] ? (uint32_t)((key[u] >> s.shift) - s.base) : 0xffffffffu;
            runs[u] = wave_runs(bucket, active[u]);
            at[u] = 0;
            if (runs[u].is_head) at[u] = bucket_begin(chunk_base, offsets, bucket) + atomicAdd(&counts[bucket], runs[u].run_length);
        }
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) {
            const uint32_t lo = __shfl((uint32_t)at[u], (int)runs[u].head_lane, 64), hi = __shfl((uint32_t)(at[u] >> 32), (int)runs[u].head_lane, 64);
            if (active[u]) alt[(((uint64_t)hi << 32) | lo) + (lane - runs[u].head_lane)] = key[u];
        }
    }
}

// sort: every bucket back into `hits`, as the reference's records, at its final place
__device__ __forceinline__ void order_sort_small(const OrderShape& s, uint64_t gwave, uint64_t nwaves) {
    const order_args_t a = order_args();
    const uint64_t* __restrict__ const alt = a->alt;
    uint64_t* __restrict__ const out = a->hits;
    uint32_t* __restrict__ const counts = a->counts;
    const uint64_t* __restrict__ const chunk_base = a->chunk_base;
    const uint32_t* __restrict__ const offsets = a->offsets;
    const uint32_t row_bits = a->row_bits;
    const uint64_t low_mask = (1ull << s.shift) - 1ull;
    // Beside the SSV kernel of the next pass a wave of this kernel is the youngest on its SIMD and would be issued last: raised
    // priority, in and out.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = fresh_lane();
    for (uint64_t b64 = gwave; b64 < s.nbuckets; b64 += nwaves) {
        const uint32_t b = (uint32_t)b64;
        const uint32_t count = uniform32(counts[b]);                               // the scatter's cursor: back at the bucket's count
        if (count == 0 || count > kSmallBucket) continue;
        if (lane == 0) counts[b] = 0;                                              // left clean for the next pass: no fill in front of it
        const uint64_t begin = bucket_begin(chunk_base, offsets, b);
        const uint64_t bucket_key = ((uint64_t)b + s.base) << s.shift;
        if (count <= 64) sort_bucket_in_a_wave<1>(alt, out, begin, count, bucket_key, low_mask, row_bits, lane);
        else if (count <= 128) sort_bucket_in_a_wave<2>(alt, out, begin, count, bucket_key, low_mask, row_bits, lane);
        else sort_bucket_in_a_wave<4>(alt, out, begin, count, bucket_key, low_mask, row_bits, lane);
    }
    __builtin_amdgcn_s_setprio(0);
}

// the rare buckets of 257 ... large_capacity records: one workgroup each, bitonic in LDS
__device__ __forceinline__ void order_sort_large(const OrderShape& s, uint32_t* lds, uint32_t nlarge, uint32_t wave) {
    if (nlarge == 0) return;
    const order_args_t a = order_args();
    const uint64_t* const alt = a->alt;
    uint64_t* const out = a->hits;
    uint32_t* const counts = a->counts;
    const uint32_t* const large_list = a->large_list;
    const uint32_t row_bits = a->row_bits;
    const uint64_t low_mask = (1ull << s.shift) - 1ull;
    const uint32_t tt = fresh_lane() + 64u * wave;
    for (uint32_t k = blockIdx.x; k < nlarge; k += gridDim.x) {
        const uint32_t b = large_list[k];
        const uint64_t begin = bucket_begin(a->chunk_base, a->offsets, b);
        const uint32_t count = counts[b];
        __syncthreads();
        if (tt == 0) counts[b] = 0;
        uint32_t N = 1;
        while (N < count) N <<= 1;
        for (uint32_t i = tt; i < N; i += kOrderThreads) lds[i] = i < count ? (uint32_t)(alt[begin + i] & low_mask) : 0xffffffffu;
        __syncthreads();
        bitonic_sort_lds<kOrderThreads>(lds, N);
        const uint64_t bucket_key = ((uint64_t)b + s.base) << s.shift;
        for (uint32_t i = tt; i < count; i += kOrderThreads) out[begin + i] = key_to_record(bucket_key | lds[i], row_bits);
        __syncthreads();
    }
}

// what the scan's outcome means for the phases behind it (wave-uniform): 0 go on, else the status to leave with
__device__ __forceinline__ uint32_t order_after_scan(uint32_t& nlarge) {
    const order_args_t a = order_args();
    OrderState* const st = a->state;
    const uint32_t oversized = uniform32(__hip_atomic_load(&st->oversized, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    nlarge = uniform32(__hip_atomic_load(&st->nlarge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t largest = uniform32(__hip_atomic_load(&st->largest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (oversized) return kOrderGeneric;                                  // (the scan left every count at zero: nothing to undo)
    if (nlarge && largest > a->large_capacity) return kOrderRetry;        // the launch's LDS does not hold the largest bucket: again, with more
    return kOrderNone;                                                    // (the counts are gone -- the scan cleared them -- but nothing was moved)
}

// ---- the kernels -------------------------------------------------------------------------------------------------------------
// count | scan | scatter | sort | finish (the large buckets and the report).
// Where passes run beside each other (an ordering stream is set: havac_ssv_set_order_stream) the ordering of pass k runs beside
// the SSV kernel of pass k + 1 and must fit into what that kernel leaves free on a SIMD -- 512 - 6 x 80 = 32 VGPRs -- or its
// workgroups wait for a whole SSV workgroup's place, at low priority, until that kernel's last round.  The three kernels that
// touch every record are held to that (amdgpu_num_vgpr counts the unified register file of gfx90a and later in pairs: 16 = 32
// registers; what does not fit -- three loop invariants of the sorter -- is parked in scratch).  The LDS sorter of the large
// buckets is not, and has a launch of its own; only its few workgroups take the ticket that finds the one which reports (a
// returning atomic per workgroup of the sort's grid -- 8,000 of them on one word -- took 0.4 ms).
__global__ __launch_bounds__(kOrderThreads) __attribute__((amdgpu_num_vgpr(16)))
void ssv_order_count(const OrderPass /* read through order_args(), phase by phase: never by name */) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u, gwave = (uint64_t)blockIdx.x * 4u + wave;
    const OrderShape s = order_shape();                 // what this pass holds, and how wide a bucket is
    if (s.status != kOrderNone) return;
    order_count(s, gwave, nwaves);
}

__global__ __launch_bounds__(kOrderThreads)
void ssv_order_scan(const OrderPass) {                  // a workgroup per kScanChunk buckets of the context's tables
    extern __shared__ uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const OrderShape s = order_shape();
    if (s.status != kOrderNone) return;
    scan_bucket_counts(lds, s.nbuckets, s.found, wave);
}

__global__ __launch_bounds__(kOrderThreads) __attribute__((amdgpu_num_vgpr(16)))
void ssv_order_scatter(const OrderPass) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u, gwave = (uint64_t)blockIdx.x * 4u + wave;
    const OrderShape s = order_shape();
    uint32_t nlarge = 0;
    if (s.status != kOrderNone || order_after_scan(nlarge) != kOrderNone) return;
    order_scatter(s, gwave, nwaves);
}

__global__ __launch_bounds__(kOrderThreads) __attribute__((amdgpu_num_vgpr(16)))
void ssv_order_sort(const OrderPass) {                  // the small buckets only; ssv_order_finish follows
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u, gwave = (uint64_t)blockIdx.x * 4u + wave;
    const OrderShape s = order_shape();
    uint32_t nlarge = 0;
    if (s.status != kOrderNone || order_after_scan(nlarge) != kOrderNone) return;
    order_sort_small(s, gwave, nwaves);
}

__global__ __launch_bounds__(kOrderThreads)
void ssv_order_finish(const OrderPass) {                // the large buckets; its last workgroup reports to the host
    extern __shared__ uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const OrderShape s = order_shape();
    if (s.status != kOrderNone) { order_leave(s, s.status, wave); return; }
    uint32_t nlarge = 0;
    if (const uint32_t how = order_after_scan(nlarge)) { order_leave(s, how, wave); return; }
    order_sort_large(s, lds, nlarge, wave);
    order_leave(s, kOrderDone, wave);
}

// ---- the check of a gathered list ---------------------------------------------------------------------------------------
// What rank 0 of a sharded run asserts about the list it has gathered before it reports it (bench.py: distributed.parity):
// the records are in the reference's emission order -- segment, then row, then column, strictly ascending, so there are no
// duplicates either (device/HavacHls.cpp:151-152,264) -- and the records rank r contributed (count[r] of them, the lists laid
// end to end in rank order) lie inside that rank's columns.  ONE grid-stride pass over the list, 16 bytes read per record
// (its own and its predecessor's, which the neighbouring lane has just fetched: L1 / L2 hits), no list-sized temporary:
// C4's gathered list is 4.46e9 records = 36 GB, more elements than a 32-bit index holds and more bytes than a second copy
// should cost on the card that already holds the receive buffers.
struct OrderReport {               // havac_order_report (include/havac_dev.h)
    unsigned long long records, out_of_order, first_out_of_order, out_of_span, first_out_of_span;
};
constexpr uint32_t kMaxCheckRanks = 1024;
__global__ __launch_bounds__(256)
void ssv_check_order(const uint64_t* __restrict__ records, uint64_t n, const uint64_t* __restrict__ rank_begin /* nranks + 1 */,
                     const uint64_t* __restrict__ span_begin, const uint64_t* __restrict__ span_end, uint32_t nranks,
                     OrderReport* __restrict__ report) {
    __shared__ uint64_t s_begin[kMaxCheckRanks + 1];
    for (uint32_t k = threadIdx.x; k <= nranks; k += blockDim.x) s_begin[k] = rank_begin[k];
    __syncthreads();
    unsigned long long bad_order = 0, bad_span = 0, first_order = ~0ull, first_span = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t rec = records[i];
        const uint64_t key = record_to_key(rec, 24);
        if (i > 0 && record_to_key(records[i - 1], 24) >= key) { bad_order++; if (i < first_order) first_order = i; }
        if (nranks) {
            // the rank whose stretch of the list holds record i: the last r with rank_begin[r] <= i
            uint32_t lo = 0, hi = nranks;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if (s_begin[mid] <= i) lo = mid; else hi = mid;
            }
            const uint64_t column = ((rec >> 14) & 0x3ffffffull) * 12288ull + (rec & 0x3fffull);
            if (column < span_begin[lo] || column >= span_end[lo]) { bad_span++; if (i < first_span) first_span = i; }
        }
    }
    if (bad_order) { atomicAdd(&report->out_of_order, bad_order); atomicMin(&report->first_out_of_order, first_order); }
    if (bad_span) { atomicAdd(&report->out_of_span, bad_span); atomicMin(&report->first_out_of_span, first_span); }
}

}  // namespace havac
