// hit_order.hip.h -- the hit queue put into the FPGA's emission order without a generic sort.
//
// The reference's device emits its records segment by segment, row by row, column by column
// (device/HavacHls.cpp:151-152,264; the sieve of device/HitReporting.cpp:178-337 keeps the columns ascending).  The SSV
// kernel here appends records in the order its 6,144 waves find them, as sort keys
//     key = segment << (14 + row_bits) | row << 14 | column-in-segment        (ssv_kernels.hip.h: hit_key)
// so device order = ascending key, and keys are unique (a cell hits at most once).  Round 2 ordered them with rocPRIM's
// radix sort: 5-6 Onesweep passes over all keys whatever their structure.  This file uses the structure:
//
//   * a BUCKET is a run of consecutive key values: (key >> shift) -- one segment, or one range of 2^k rows of a
//     segment when segments hold many records (tall collections).  Buckets in ascending order concatenate to the
//     sorted whole, so only the inside of a bucket has to be sorted;
//   * ssv_bucket_count: one pass over the keys, one non-returning atomic per RUN of equal buckets inside a wave (a
//     burst of the SSV kernel's queue comes from one tile: runs are long);
//   * ssv_bucket_scan: exclusive scan of the counts (one launch: a workgroup per 2048 buckets, the last one to finish
//     scans the chunk totals), the list of buckets too big for the small sorter, and the `oversized` flag when a bucket
//     exceeds what an LDS sort holds;
//   * ssv_bucket_scatter: second pass over the keys, each run to its bucket's stretch of a second buffer (one returning
//     atomic per run);
//   * ssv_bucket_sort_small: one WAVE per bucket of up to 256 records sorts the keys' low bits (32-bit) in registers --
//     a bitonic network whose short-distance stages are cross-lane reads and whose long-distance stages are register
//     exchanges, no LDS and no barrier -- and writes the bucket back as the reference's packed RECORDS
//     (device/HitReporting.cpp:421-430) at its final place: the separate key -> record pass of round 2 is gone;
//     ssv_bucket_sort_large: the rare buckets of 257 ... 16,384 records, one workgroup each, bitonic in LDS.
//
// Four passes over the data in all (count, scatter read + write, sort read + write) against 11+ for the radix sort.
// A bucket holds at most 2^(shift-14) rows x 12288 columns of cells, but nothing bounds how many of them hit: when
// one bucket exceeds kLargeBucket records (a clump 16x denser than the launch's average, e.g. one low-complexity model
// in a sparse collection) the scan raises `oversized`, scatter and sort do nothing, and the host falls back to the
// radix sort for that pass (havac_dev.hip, order_records).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "ssv_kernels.hip.h"

namespace havac {

constexpr uint32_t kSmallBucket = 256;       // records one WAVE sorts in registers (4 per lane)
constexpr uint32_t kLargeBucket = 16384;     // records the large sorter holds (64 KB of LDS, one workgroup)
constexpr uint32_t kTargetBucket = 128;      // average records per bucket the host aims for when it picks the shift
constexpr uint32_t kScanChunk = 2048;        // buckets one workgroup of the scan takes (256 threads x 8)

struct OrderState {                // device words the kernels share (one cache line)
    uint32_t oversized;            // a bucket exceeds kLargeBucket: nothing was moved, the host takes the generic path
    uint32_t nlarge;               // buckets with kSmallBucket < count <= kLargeBucket, listed in large_list
    uint32_t largest;              // the largest bucket (reporting)
    uint32_t chunks_done;          // scan: workgroups that have written their chunk's total (the last one scans the totals)
};

// runs of equal bucket numbers among the active lanes of a wave (active lanes are a prefix of the wave)
struct WaveRuns { unsigned long long heads; uint32_t head_lane, run_length; bool is_head; };
__device__ __forceinline__ WaveRuns wave_runs(uint32_t bucket, bool active) {
    const uint32_t lane = __lane_id();
    const uint32_t before = __shfl_up(bucket, 1, 64);
    WaveRuns r;
    r.is_head = active && (lane == 0 || before != bucket);
    r.heads = __ballot(r.is_head);
    const unsigned long long actives = __ballot(active);
    const uint32_t nactive = (uint32_t)__popcll(actives);
    const unsigned long long upto_me = r.heads & ((2ull << lane) - 1ull);                // heads at or before this lane
    r.head_lane = upto_me ? 63u - (uint32_t)__clzll(upto_me) : 0u;
    const unsigned long long after_head = r.heads & ~((2ull << r.head_lane) - 1ull);
    const uint32_t run_end = after_head ? (uint32_t)__builtin_ctzll(after_head) : nactive;
    r.run_length = run_end - r.head_lane;
    return r;
}

// (count and scatter: a wave takes 256 consecutive keys at a time, four coalesced loads in flight per lane before the first
// is used; few, long-lived workgroups instead of one short one per 256 keys -- next to the SSV kernel of the following pass
// what an ordering kernel costs is the TIME its workgroups hold a CU's wave slots and registers, mostly waiting for memory)
constexpr int kKeysPerLane = 4;              // count
constexpr int kScatterKeysPerLane = 2;       // scatter (more would cost it the 32-VGPR budget, see ssv_bucket_sort_small)

__global__ __launch_bounds__(256)
void ssv_bucket_count(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint64_t base, uint32_t* __restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4u;
    const uint64_t span = 64u * kKeysPerLane;
    for (uint64_t first = wave * span; first < n; first += nwaves * span) {
        uint64_t key[kKeysPerLane];
#pragma unroll
        for (int u = 0; u < kKeysPerLane; u++) { const uint64_t i = first + 64u * u + lane; key[u] = i < n ? keys[i] : 0; }
#pragma unroll
        for (int u = 0; u < kKeysPerLane; u++) {
            const bool active = first + 64u * u + lane < n;
            const uint32_t bucket = active ? (uint32_t)((key[u] >> shift) - base) : 0xffffffffu;
            const WaveRuns r = wave_runs(bucket, active);
            if (r.is_head) atomicAdd(&counts[bucket], r.run_length);
        }
    }
}

// Exclusive scan of the bucket counts.  Workgroup c takes buckets [c * kScanChunk, (c+1) * kScanChunk): it writes each
// bucket's offset INSIDE the chunk (local_offsets), lists the chunk's large buckets, clears the counts (the scatter uses
// them as cursors and leaves the counts in them again) and notes the chunk's total; the workgroup that finishes LAST scans
// the totals into chunk_base (64-bit: a pass may hold more than 2^32 records).  A bucket begins at
// chunk_base[b / kScanChunk] + local_offsets[b].  One launch whatever the number of buckets.
__global__ __launch_bounds__(256)
void ssv_bucket_scan(uint32_t* __restrict__ counts, uint32_t nbuckets, uint32_t* __restrict__ local_offsets, uint64_t* __restrict__ chunk_base,
                     uint32_t nchunks, uint32_t* __restrict__ large_list, OrderState* __restrict__ state, uint64_t nrecords) {
    __shared__ uint32_t tile[kScanChunk];
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t last_block;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t first = blockIdx.x * kScanChunk;
    uint32_t most = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {                       // coalesced: thread t takes buckets first + t + 256 i
        const uint32_t b = first + t + 256u * i;
        const uint32_t c = b < nbuckets ? counts[b] : 0u;
        tile[t + 256u * i] = c;
        most = c > most ? c : most;
        if (b < nbuckets) {
            counts[b] = 0;
            if (c > kSmallBucket && c <= kLargeBucket) large_list[atomicAdd(&state->nlarge, 1u)] = b;
        }
    }
    __syncthreads();
    uint32_t mine[8], sum = 0;                          // thread t scans buckets first + 8 t ... 8 t + 7
#pragma unroll
    for (int i = 0; i < 8; i++) { mine[i] = sum; sum += tile[8u * t + i]; }
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
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) tile[8u * t + i] = before + mine[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t b = first + t + 256u * i;
        if (b < nbuckets) local_offsets[b] = tile[t + 256u * i];
    }
    // the largest bucket of the chunk
#pragma unroll
    for (int d = 32; d >= 1; d /= 2) { const uint32_t other = __shfl_xor(most, d, 64); most = other > most ? other : most; }
    if (lane == 0 && most) {
        atomicMax(&state->largest, most);
        if (most > kLargeBucket) atomicOr(&state->oversized, 1u);
    }
    // this chunk's total; the last workgroup to get here scans the totals
    if (t == 255) {
        chunk_base[blockIdx.x] = (uint64_t)(before + sum);          // a total for now, a base after the scan below
        __threadfence();
        last_block = atomicAdd(&state->chunks_done, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_block) return;
    __threadfence();
    __shared__ uint64_t part[256];
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
        if (c < nchunks) { const uint64_t mine_total = chunk_base[c]; chunk_base[c] = at; at += mine_total; }
    }
    if (t == 255) {
        chunk_base[nchunks] = part[255];
        state->chunks_done = 0;
        // The counts must add up to the records of this pass.  They do unless a pass before this one was cut short (a HIP error
        // between its kernels) and left counts behind: then nothing is moved -- offsets from such counts would point outside the
        // buffers -- and the host clears the counts and orders this pass with the radix sort (`oversized` = 2).
        if (part[255] != nrecords) atomicOr(&state->oversized, 2u);
    }
}

__device__ __forceinline__ uint64_t bucket_begin(const uint64_t* __restrict__ chunk_base, const uint32_t* __restrict__ local_offsets, uint32_t b) {
    return chunk_base[b / kScanChunk] + local_offsets[b];
}

__global__ __launch_bounds__(256)
void ssv_bucket_scatter(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint64_t base, const uint64_t* __restrict__ chunk_base,
                        const uint32_t* __restrict__ local_offsets, uint32_t* __restrict__ cursors, uint64_t* __restrict__ out,
                        const OrderState* __restrict__ state) {
    if (state->oversized) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4u;
    const uint64_t span = 64u * kScatterKeysPerLane;
    for (uint64_t first = wave * span; first < n; first += nwaves * span) {
        uint64_t key[kScatterKeysPerLane];
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) { const uint64_t i = first + 64u * u + lane; key[u] = i < n ? keys[i] : 0; }
        uint64_t at[kScatterKeysPerLane];
        WaveRuns runs[kScatterKeysPerLane];
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) {          // the four returning atomics of a lane's heads are in flight together
            const bool active = first + 64u * u + lane < n;
            const uint32_t bucket = active ? (uint32_t)((key[u] >> shift) - base) : 0xffffffffu;
            runs[u] = wave_runs(bucket, active);
            at[u] = 0;
            if (runs[u].is_head) at[u] = bucket_begin(chunk_base, local_offsets, bucket) + atomicAdd(&cursors[bucket], runs[u].run_length);
        }
#pragma unroll
        for (int u = 0; u < kScatterKeysPerLane; u++) {
            const uint32_t lo = __shfl((uint32_t)at[u], (int)runs[u].head_lane, 64), hi = __shfl((uint32_t)(at[u] >> 32), (int)runs[u].head_lane, 64);
            if (first + 64u * u + lane < n) out[(((uint64_t)hi << 32) | lo) + (lane - runs[u].head_lane)] = key[u];
        }
    }
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

// Held to 32 VGPRs (four registers of keys per lane): that is what the SSV kernel of the FOLLOWING pass leaves free on a SIMD
// (6 waves x 80 of 512 registers), so a wave of this kernel runs in that kernel's shadow instead of taking the place of one of
// its waves.  The host picks the buckets' size so that nearly all of them come here (kTargetBucket on average, C2: 124).
__global__ __launch_bounds__(256)
void ssv_bucket_sort_small(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const uint64_t* __restrict__ chunk_base,
                           const uint32_t* __restrict__ local_offsets, uint32_t* __restrict__ counts, uint32_t nbuckets,
                           uint32_t shift, uint64_t base, uint32_t row_bits, const OrderState* __restrict__ state) {
    if (state->oversized) return;
    // Beside the SSV kernel of the next pass a wave of this kernel is the youngest on its SIMD and would be issued last: it
    // would sit in its wave slot -- and keep a whole SSV workgroup from starting on the CU -- several times longer than its
    // few hundred instructions need.  Raised priority: in and out.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t low_mask = (1ull << shift) - 1ull;
    for (uint64_t b64 = (uint64_t)blockIdx.x * 4u + wave; b64 < nbuckets; b64 += (uint64_t)gridDim.x * 4u) {
        const uint32_t b = (uint32_t)b64;
        const uint32_t count = __builtin_amdgcn_readfirstlane(counts[b]);       // the scatter's cursor: back at the bucket's count
        if (count == 0 || count > kSmallBucket) continue;
        if (lane == 0) counts[b] = 0;                                            // left clean for the next pass: no fill in front of it
        const uint64_t begin = bucket_begin(chunk_base, local_offsets, b);
        const uint64_t bucket_key = ((uint64_t)b + base) << shift;
        if (count <= 64) sort_bucket_in_a_wave<1>(in, out, begin, count, bucket_key, low_mask, row_bits, lane);
        else if (count <= 128) sort_bucket_in_a_wave<2>(in, out, begin, count, bucket_key, low_mask, row_bits, lane);
        else sort_bucket_in_a_wave<4>(in, out, begin, count, bucket_key, low_mask, row_bits, lane);
    }
}

// ---- large buckets: one workgroup, bitonic in LDS ---------------------------------------------------------------------------
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

// (256 threads: next to the SSV kernel of the following pass a workgroup of this size fits into what that kernel leaves free
// of a CU -- 8 of 32 wave slots, 76 of 160 KB of LDS -- where a 1024-thread workgroup would have the CU drained for it)
__global__ __launch_bounds__(256)
void ssv_bucket_sort_large(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const uint64_t* __restrict__ chunk_base,
                           const uint32_t* __restrict__ local_offsets, uint32_t* __restrict__ counts,
                           const uint32_t* __restrict__ large_list, uint32_t shift, uint64_t base, uint32_t row_bits,
                           const OrderState* __restrict__ state, OrderState* __restrict__ next_state, OrderState* __restrict__ host_state) {
    __shared__ uint32_t v[kLargeBucket];
    // The last kernel of a pass's ordering.  Workgroup 0 tells the host what happened (a pinned word the host reads after the
    // stream's event: no copy behind this kernel) and clears the state words of the NEXT pass (two sets, used alternately:
    // no fill in front of the next pass either).
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(&host_state->largest, state->largest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_state->nlarge, state->nlarge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_state->oversized, state->oversized, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        next_state->oversized = 0; next_state->nlarge = 0; next_state->largest = 0; next_state->chunks_done = 0;
    }
    if (state->oversized) return;
    const uint32_t nlarge = state->nlarge;
    const uint64_t low_mask = (1ull << shift) - 1ull;
    for (uint32_t k = blockIdx.x; k < nlarge; k += gridDim.x) {
        const uint32_t b = large_list[k];
        const uint64_t begin = bucket_begin(chunk_base, local_offsets, b);
        const uint32_t count = counts[b];
        __syncthreads();
        if (threadIdx.x == 0) counts[b] = 0;
        uint32_t N = 1;
        while (N < count) N <<= 1;
        for (uint32_t i = threadIdx.x; i < N; i += 256) v[i] = i < count ? (uint32_t)(in[begin + i] & low_mask) : 0xffffffffu;
        __syncthreads();
        bitonic_sort_lds<256>(v, N);
        const uint64_t bucket_key = ((uint64_t)b + base) << shift;
        for (uint32_t i = threadIdx.x; i < count; i += 256) out[begin + i] = key_to_record(bucket_key | v[i], row_bits);
        __syncthreads();
    }
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
