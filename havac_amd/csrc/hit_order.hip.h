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
//   * ssv_bucket_scan: exclusive scan of the counts (one workgroup), the list of buckets too big for the small sorter,
//     and the `oversized` flag when a bucket exceeds what an LDS sort holds;
//   * ssv_bucket_scatter: second pass over the keys, each run to its bucket's stretch of a second buffer (one returning
//     atomic per run);
//   * ssv_bucket_sort_small / _large: one workgroup per bucket sorts its keys' low bits (32-bit, in LDS, bitonic) and
//     writes the bucket back as the reference's packed RECORDS (device/HitReporting.cpp:421-430) at its final place --
//     the separate key -> record pass of round 2 is gone.
//
// Four passes over the data in all (count, scatter read + write, sort read + write) against 11+ for the radix sort.
// A bucket holds at most 2^(shift-14) rows x 12288 columns of cells, but nothing bounds how many of them hit: when
// one bucket exceeds kLargeBucket records (a clump 16x denser than the launch's average, e.g. one low-complexity model
// in a sparse collection) the scan raises `oversized`, scatter and sort do nothing, and the host falls back to the
// radix sort for that pass (havac_dev.hip, order_records).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssv_kernels.hip.h"

namespace havac {

constexpr uint32_t kSmallBucket = 2048;      // records the small sorter holds (8 KB of LDS, 256 threads)
constexpr uint32_t kLargeBucket = 16384;     // records the large sorter holds (64 KB of LDS, 1024 threads)
constexpr uint32_t kTargetBucket = 512;      // average records per bucket the host aims for when it picks the shift

struct OrderState {                // device words the kernels share (one cache line)
    uint32_t oversized;            // a bucket exceeds kLargeBucket: nothing was moved, the host takes the generic path
    uint32_t nlarge;               // buckets with kSmallBucket < count <= kLargeBucket, listed in large_list
    uint32_t largest;              // the largest bucket (reporting)
    uint32_t pad;
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

__global__ __launch_bounds__(256)
void ssv_bucket_count(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint64_t base, uint32_t* __restrict__ counts) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;               // every wave runs the same number of rounds (the ballots need whole waves)
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t k = 0; k < rounds; k++, i += stride) {
        const bool active = i < n;
        const uint32_t bucket = active ? (uint32_t)((keys[i] >> shift) - base) : 0xffffffffu;
        const WaveRuns r = wave_runs(bucket, active);
        if (r.is_head) atomicAdd(&counts[bucket], r.run_length);
    }
}

// One workgroup: offsets[b] = records in buckets before b (64-bit: a pass may hold more than 2^32 records),
// offsets[nbuckets] = n; counts[] are cleared (the scatter uses them as cursors).
__global__ __launch_bounds__(1024)
void ssv_bucket_scan(uint32_t* __restrict__ counts, uint32_t nbuckets, uint64_t* __restrict__ offsets, uint32_t* __restrict__ large_list,
                     OrderState* __restrict__ state) {
    __shared__ uint64_t partial[1024];
    __shared__ uint32_t biggest[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nbuckets + 1023u) / 1024u;
    const uint64_t first = (uint64_t)t * per;
    uint64_t sum = 0;
    uint32_t most = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint64_t b = first + k;
        if (b < nbuckets) { const uint32_t c = counts[b]; sum += c; most = c > most ? c : most; }
    }
    partial[t] = sum; biggest[t] = most;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d *= 2) {           // inclusive scan of the 1024 partial sums
        const uint64_t add = t >= d ? partial[t - d] : 0;
        const uint32_t other = t >= d ? biggest[t - d] : 0;
        __syncthreads();
        partial[t] += add; biggest[t] = other > biggest[t] ? other : biggest[t];
        __syncthreads();
    }
    uint64_t at = partial[t] - sum;
    const bool oversized = biggest[1023] > kLargeBucket;
    for (uint32_t k = 0; k < per; k++) {
        const uint64_t b = first + k;
        if (b < nbuckets) {
            const uint32_t c = counts[b];
            offsets[b] = at;
            at += c;
            counts[b] = 0;
            if (!oversized && c > kSmallBucket) large_list[atomicAdd(&state->nlarge, 1u)] = (uint32_t)b;
        }
    }
    if (t == 1023) { offsets[nbuckets] = partial[1023]; state->oversized = oversized ? 1u : 0u; state->largest = biggest[1023]; }
}

__global__ __launch_bounds__(256)
void ssv_bucket_scatter(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint64_t base, const uint64_t* __restrict__ offsets,
                        uint32_t* __restrict__ cursors, uint64_t* __restrict__ out, const OrderState* __restrict__ state) {
    if (state->oversized) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = __lane_id();
    for (uint64_t k = 0; k < rounds; k++, i += stride) {
        const bool active = i < n;
        const uint64_t key = active ? keys[i] : 0;
        const uint32_t bucket = active ? (uint32_t)((key >> shift) - base) : 0xffffffffu;
        const WaveRuns r = wave_runs(bucket, active);
        uint64_t at = 0;
        if (r.is_head) at = offsets[bucket] + atomicAdd(&cursors[bucket], r.run_length);
        const uint32_t lo = __shfl((uint32_t)at, (int)r.head_lane, 64), hi = __shfl((uint32_t)(at >> 32), (int)r.head_lane, 64);
        if (active) out[(((uint64_t)hi << 32) | lo) + (lane - r.head_lane)] = key;
    }
}

// bitonic sort of N (a power of two) 32-bit values in LDS by T threads
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

// One bucket: `count` keys at in[begin ..] -> sorted, as records, at out[begin ..].
template <uint32_t T, uint32_t CAP>
__device__ __forceinline__ void sort_bucket(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t begin, uint32_t count,
                                            uint64_t bucket_key /* (bucket + base) << shift */, uint32_t shift, uint32_t row_bits, uint32_t* v) {
    uint32_t N = 1;
    while (N < count) N <<= 1;
    const uint64_t low_mask = (shift >= 64) ? ~0ull : ((1ull << shift) - 1ull);
    for (uint32_t i = threadIdx.x; i < N; i += T) v[i] = i < count ? (uint32_t)(in[begin + i] & low_mask) : 0xffffffffu;
    __syncthreads();
    if (N > 1) bitonic_sort_lds<T>(v, N);
    for (uint32_t i = threadIdx.x; i < count; i += T) out[begin + i] = key_to_record(bucket_key | v[i], row_bits);
}

__global__ __launch_bounds__(256)
void ssv_bucket_sort_small(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const uint64_t* __restrict__ offsets, uint32_t nbuckets,
                           uint32_t shift, uint64_t base, uint32_t row_bits, const OrderState* __restrict__ state) {
    __shared__ uint32_t v[kSmallBucket];
    if (state->oversized) return;
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {      // (a launch holds at most 2^31-1 workgroups per dimension)
        const uint64_t begin = offsets[b];
        const uint64_t count = offsets[b + 1] - begin;
        if (count == 0 || count > kSmallBucket) continue;
        sort_bucket<256, kSmallBucket>(in, out, begin, (uint32_t)count, ((uint64_t)b + base) << shift, shift, row_bits, v);
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024)
void ssv_bucket_sort_large(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const uint64_t* __restrict__ offsets,
                           const uint32_t* __restrict__ large_list, uint32_t shift, uint64_t base, uint32_t row_bits,
                           const OrderState* __restrict__ state) {
    __shared__ uint32_t v[kLargeBucket];
    if (state->oversized) return;
    const uint32_t nlarge = state->nlarge;
    for (uint32_t k = blockIdx.x; k < nlarge; k += gridDim.x) {
        const uint32_t b = large_list[k];
        const uint64_t begin = offsets[b];
        const uint32_t count = (uint32_t)(offsets[b + 1] - begin);
        sort_bucket<1024, kLargeBucket>(in, out, begin, count, ((uint64_t)b + base) << shift, shift, row_bits, v);
        __syncthreads();
    }
}

}  // namespace havac
