// ssv_kernels.hip.h -- CDNA4 (gfx950) kernels of the SSV filter.
//
// What is computed (reference: device/HavacHls.cpp:220-402, oracle form
// test/softSsv/SoftSsv.cpp:31-62):
//     S[p][s] = S[p-1][s-1] + M[p][seq[s]]      (0 on row 0 / column 0)
//     S < 0 -> 0 ;  S >= 256 -> report (p, s), S = 0
//
// How it is laid out here (nothing like the FPGA's systolic row of 12288 PEs):
//
//  * DIAGONAL-STATIONARY.  A cell depends only on its own diagonal d = s - p,
//    so a lane owns 32 adjacent diagonals for the whole height of the matrix
//    and its scores never move.  No inter-lane shuffle, no inter-tile carry
//    (the FPGA's scoreQueue, device/HavacHls.cpp:451-465, disappears) and no
//    halo between GPUs.  A wave owns 64 x 32 = 2048 diagonals (a "tile").
//  * TWO CELLS PER VGPR as packed int16 holding (score - 32768): one
//    v_pk_add_i16 with clamp is add + clamp-at-zero, and a crossing of 256 is
//    bit 8 of the unbiased score (scores never exceed 255 + 127).
//  * MATCH SCORES BY v_perm_b32.  The model row is widened once to four int16
//    (8 bytes, {A,C | G,T}); the 2-bit symbols of a diagonal pair are expanded
//    once per 32-row chunk into a byte selector [2a,2a+1,2b,2b+1], so one
//    v_perm_b32 yields both sign-extended match scores.  The symbol window
//    slides one position per row; even rows use the aligned selector words,
//    odd rows the words shifted by 16 bits (v_alignbit), both indexed
//    statically in the fully unrolled 32-row chunk, so the slide costs nothing.
//  * OUTSIDE THE MATRIX (columns < 0 or >= N, rows >= nrows) the selector
//    / row yields a large negative score, which pins the cell at 0 and can
//    never hit: no per-cell predicate anywhere in the hot loop.
//  * HITS are rare (about 1e-5 per cell on Dfam-like models): the 16 score
//    registers are OR-ed, one wave-wide test per row finds bit 8, and only then
//    the slow path emits and resets.  Records are appended through one atomic
//    per wave-row and ordered afterwards (ssv_order_* below).
//
// Roofline: integer VALU issue; HBM traffic is
// N/4 + 8*rows bytes per tile sweep, thousands of cells per byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace havac {

constexpr int kDiagsPerLane = 32;                 // 16 VGPRs x 2 int16
constexpr int kRegs = kDiagsPerLane / 2;
constexpr int kTileDiags = 64 * kDiagsPerLane;    // one wave
constexpr int kChunkRows = 32;                    // rows per unrolled chunk
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kScoreZero = 0x80008000u;      // two cells at score 0 (bias -32768)
constexpr uint32_t kHitBits = 0x01000100u;        // bit 8 of either unbiased score
constexpr uint32_t kPadSelector = 0x0d0c0d0cu;    // v_perm: bytes (0x00,0xFF) -> int16 -256 twice

// sort key of a hit: (segment, row, column-in-segment) -- the FPGA's emission
// order (device/HavacHls.cpp:151-152,264; device/HitReporting.cpp:178-337)
__device__ __forceinline__ uint64_t hit_key(uint32_t row, uint64_t column) {
    uint64_t seg = column / 12288u;
    uint64_t in_seg = column - seg * 12288u;
    return (seg << 38) | ((uint64_t)row << 14) | in_seg;
}

// key -> the reference's packed record (device/HitReporting.cpp:421-430)
__device__ __forceinline__ uint64_t key_to_record(uint64_t key) {
    uint64_t in_seg = key & 0x3fffull;
    uint64_t row = (key >> 14) & 0xffffffull;
    uint64_t seg = key >> 38;
    return in_seg | (seg << 14) | (row << 40);
}
__device__ __forceinline__ uint64_t record_to_key(uint64_t rec) {
    uint64_t in_seg = rec & 0x3fffull;
    uint64_t seg = (rec >> 14) & 0x3ffffffull;
    uint64_t row = rec >> 40;
    return (seg << 38) | (row << 14) | in_seg;
}

// ---------------------------------------------------------------------------
// model int8 [row][A,C,G,T]  ->  int16 x4 per row, padded with rows of -128 up
// to a whole number of chunks (a padding row can only lower a score).
// Replaces nothing in the reference: the FPGA muxes bytes directly
// (device/HavacHls.cpp:429-442); this is the layout v_perm_b32 wants.
__global__ void ssv_expand_model(const int8_t* __restrict__ phmm, uint32_t nrows,
                                 uint2* __restrict__ rows16, uint32_t nrows_padded) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows_padded) return;
    int a = -128, c = -128, g = -128, t = -128;
    if (r < nrows) {
        uint32_t w = reinterpret_cast<const uint32_t*>(phmm)[r];
        a = (int8_t)(w & 0xff); c = (int8_t)((w >> 8) & 0xff);
        g = (int8_t)((w >> 16) & 0xff); t = (int8_t)(w >> 24);
    }
    uint2 o;
    o.x = ((uint32_t)a & 0xffffu) | ((uint32_t)c << 16);
    o.y = ((uint32_t)g & 0xffffu) | ((uint32_t)t << 16);
    rows16[r] = o;
}

// ---------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_add_pk16(uint32_t a, uint32_t b) {
    short2v r = __builtin_elementwise_add_sat(__builtin_bit_cast(short2v, a),
                                              __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}

// selector word for the symbol pair in bits [3:0] of `nib`: bytes
// [2a, 2a+1, 2b, 2b+1] (a = low symbol)
__device__ __forceinline__ uint32_t pair_selector(uint32_t w, int shift) {
    uint32_t a = (w >> shift) & 3u;
    uint32_t b = (w >> (shift + 2)) & 3u;
    return ((b * 0x0202u) << 16) + (a * 0x0202u + 0x01000100u);   // 24-bit multiplies + v_lshl_add
}

// 8 bytes = 32 symbols at symbol offset `pos` (multiple of 32); zero outside the buffer
__device__ __forceinline__ uint2 load_symbols(const uint8_t* __restrict__ seq, int64_t nsymbols, int64_t pos,
                                              bool checked) {
    if (checked && (pos < 0 || pos + 32 > nsymbols)) return make_uint2(0u, 0u);
    return *reinterpret_cast<const uint2*>(seq + (pos >> 2));
}

// ---- hit queue --------------------------------------------------------------
// Counterpart of the FPGA's five-stage hit sieve (device/HitReporting.cpp:12-417).
// Hot loop: the 16 score registers of a row are OR-ed and one wave-wide test
// looks for bit 8.  Only a wave that sees it runs the per-row slow path, which
// turns the crossings into one 32-bit mask per lane (bit b = diagonal b of the
// lane), resets those cells to 0 and parks the mask in LDS.  Once per 32-row
// chunk the parked masks are turned into records, staged in the wave's LDS
// slice and appended to the global queue in bursts: one returning atomic per
// burst instead of one per hit (a single counter word sustains only ~90
// returning atomics per microsecond chip-wide, which capped the first version
// of this kernel at ~90 M hits/s).
constexpr int kHitStage = 128;                    // records staged per wave (1 KiB of LDS)

struct HitSink {
    uint64_t* hits;                // global queue of sort keys (hit_key)
    unsigned long long* hit_count; // records found so far (may run past capacity)
    uint64_t hit_capacity;
    uint64_t* stage;               // this wave's LDS slice, kHitStage records
    uint32_t* row_masks;           // this wave's LDS slice, kChunkRows x 64 masks
};

__device__ __forceinline__ uint32_t flush_hits(const HitSink& sink, uint32_t staged, int lane) {
    if (staged == 0) return 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(sink.hit_count, (unsigned long long)staged);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    for (uint32_t i = lane; i < staged; i += 64) {
        const unsigned long long idx = base + i;
        if (idx < sink.hit_capacity) sink.hits[idx] = sink.stage[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return 0;
}

// Once per chunk, only if some row of the chunk parked masks.  `rows_with_hits`
// has bit r set for row p0 + r.  The column of bit b of lane l on row p0 + r is
// wave_column0 + r + 32*l + b.  Returns the new number of staged records.
__device__ __noinline__ uint32_t drain_rows(const HitSink sink, uint32_t staged, uint32_t rows_with_hits, uint32_t p0,
                                            int64_t wave_column0, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (rows_with_hits) {
        const int r = __builtin_ctz(rows_with_hits);
        rows_with_hits &= rows_with_hits - 1;
        const uint32_t mask = sink.row_masks[r * 64 + lane];
        unsigned long long lanes = __ballot(mask != 0);
        while (lanes) {
            const int src = __builtin_ctzll(lanes);
            lanes &= lanes - 1;
            const uint32_t m = __builtin_amdgcn_readlane(mask, src);
            if (lane < 32 && ((m >> lane) & 1u)) {
                const uint32_t pos = staged + __popc(m & ((1u << lane) - 1u));
                sink.stage[pos] = hit_key(p0 + r, (uint64_t)(wave_column0 + r + 32 * src + lane));
            }
            staged += __popc(m);
            if (staged > kHitStage - 32) staged = flush_hits(sink, staged, lane);
        }
    }
    return staged;
}

// The expanded model is read through the constant address space: the address
// is wave-uniform, so hipcc emits s_load_dwordx* into SGPRs (no VGPR, no
// vmcnt wait in the row loop).  Legal because the kernel never writes it.
typedef uint32_t row16_t __attribute__((ext_vector_type(2)));   // {A,C} , {G,T} as int16 pairs
typedef const __attribute__((address_space(4))) row16_t* const_rows_t;

// One model row over the lane's 32 diagonals.  R is the row inside the chunk:
// a template parameter so that every selector index is a compile-time constant
// and the sliding symbol window costs no instruction.
template <int R>
__device__ __forceinline__ void row_step(uint32_t (&x)[kRegs], const uint32_t (&W)[32], const uint32_t (&Wodd)[31],
                                         const row16_t row, uint32_t* __restrict__ row_masks, uint32_t& rows_with_hits,
                                         int lane) {
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < kRegs; i++) {
        const uint32_t sel = (R & 1) ? Wodd[(R >> 1) + i] : W[(R >> 1) + i];
        const uint32_t m = __builtin_amdgcn_perm(row.y, row.x, sel);
        x[i] = sat_add_pk16(x[i], m);
        any |= x[i];
    }
    if (__builtin_expect(__any((any & kHitBits) != 0), 0)) {
        uint32_t mask = 0;
#pragma unroll
        for (int i = 0; i < kRegs; i++) {
            const uint32_t h = (x[i] >> 8) & 0x00010001u;              // bit 0 / bit 16: low / high cell crossed 256
            mask |= ((h | (h >> 15)) & 3u) << (2 * i);
            const uint32_t sel16 = h * 0xffffu;                       // 0xffff over each crossed cell
            x[i] = (x[i] & ~sel16) | (kScoreZero & sel16);             // crossed cells restart at 0 (SoftSsv.cpp:43-44)
        }
        row_masks[R * 64 + lane] = mask;
        rows_with_hits |= 1u << R;
    }
}

template <int... R>
__device__ __forceinline__ void chunk_rows(uint32_t (&x)[kRegs], const uint32_t (&W)[32], const uint32_t (&Wodd)[31],
                                           const row16_t (&row)[kChunkRows], uint32_t* __restrict__ row_masks,
                                           uint32_t& rows_with_hits, int lane, std::integer_sequence<int, R...>) {
    (row_step<R>(x, W, Wodd, row[R], row_masks, rows_with_hits, lane), ...);
}

// Selector words of the 4 symbols of one packed byte, through a 256-entry LDS
// table (2 KiB per workgroup, built once): two words per ds_read_b64 instead
// of ten VALU instructions.
__device__ __forceinline__ uint2 byte_selectors(const uint2* __restrict__ lut, uint32_t w, int byte) {
    const uint32_t off = (byte == 0 ? (w << 3) : (w >> (8 * byte - 3))) & 0x7f8u;   // byte value * 8
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(lut) + off);
}

__global__ __launch_bounds__(64 * kWavesPerBlock)
void ssv_diag_kernel(const uint8_t* __restrict__ seq, const int64_t nsymbols, const uint2* __restrict__ rows16,
                     const uint32_t nrows_padded, const int64_t first_diag, const uint32_t tile_begin,
                     const uint32_t tile_end, uint64_t* __restrict__ hits, unsigned long long* __restrict__ hit_count,
                     const uint64_t hit_capacity, const uint32_t* __restrict__ abort_flag) {
    __shared__ uint64_t hit_stage[kWavesPerBlock][kHitStage];
    __shared__ uint32_t hit_masks[kWavesPerBlock][kChunkRows * 64];
    __shared__ uint2 selector_lut[256];

    // table entry b: selectors of symbol pairs (s0,s1) and (s2,s3) of packed byte b
    selector_lut[threadIdx.x] = make_uint2(pair_selector(threadIdx.x, 0), pair_selector(threadIdx.x, 4));
    __syncthreads();

    const int lane = threadIdx.x & 63;
    // readfirstlane: everything derived from the tile index is wave-uniform, which lets hipcc keep
    // the row pointer in SGPRs (scalar loads of the model rows) and branch on `edge` with SALU
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = tile_begin + blockIdx.x * kWavesPerBlock + wave;
    if (tile >= tile_end) return;
    const HitSink sink{hits, hit_count, hit_capacity, hit_stage[wave], hit_masks[wave]};
    uint32_t staged = 0;          // wave-uniform

    const int64_t d0 = first_diag + (int64_t)tile * kTileDiags;     // wave's first diagonal
    const int64_t dl = d0 + lane * kDiagsPerLane;                   // lane's first diagonal
    // rows whose cells of this tile can lie inside the matrix
    int64_t p_lo = -d0 - kTileDiags;                                // first chunk touching column >= 0
    if (p_lo < 0) p_lo = 0;
    int64_t p_hi = nsymbols - d0;                                   // first chunk entirely at columns >= N
    if (p_hi > (int64_t)nrows_padded) p_hi = nrows_padded;
    if (p_lo >= p_hi) return;

    uint32_t x[kRegs];
#pragma unroll
    for (int i = 0; i < kRegs; i++) x[i] = kScoreZero;

    // Selector window of the current chunk, symbol positions [j, j+64) with j = dl + p0:
    // W[k] serves symbols (j+2k, j+2k+1), Wodd[k] symbols (j+2k+1, j+2k+2).  The upper half of one
    // chunk's window is the lower half of the next, so each chunk expands only 32 new symbols.
    uint32_t W[32], Wodd[31];
    auto expand = [&](int64_t rel, int base) {     // symbols [dl+rel, dl+rel+32) -> W[base .. base+16); rel is wave-uniform
        const bool edge = (d0 + rel < 0) || (d0 + rel + kTileDiags > nsymbols);   // the wave's 2048 positions
        const int64_t pos = dl + rel;
        const uint2 packed = load_symbols(seq, nsymbols, pos, edge);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint2 s0 = byte_selectors(selector_lut, packed.x, b);
            const uint2 s1 = byte_selectors(selector_lut, packed.y, b);
            W[base + 2 * b] = s0.x;     W[base + 2 * b + 1] = s0.y;
            W[base + 8 + 2 * b] = s1.x; W[base + 8 + 2 * b + 1] = s1.y;
        }
        if (edge) {
            // positions outside [0, N) score -256: pins the cell at 0, never hits
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int64_t q = pos + 2 * k;     // q and q+1 are in or out together (pos, N even)
                if (q < 0 || q >= nsymbols) W[base + k] = kPadSelector;
            }
        }
    };
    expand(p_lo, 16);

    for (int64_t p0 = p_lo; p0 < p_hi; p0 += kChunkRows) {
        // abort: a device word, read past the caches every 2048 rows (never on a wave's first chunk, so
        // short models pay nothing)
        if (abort_flag && ((p0 & 2047) == 0) && p0 != p_lo &&
            __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        // rows of the chunk: fetched in two batches of scalar loads, the second lands while rows 0..15 run
        const const_rows_t rows = (const_rows_t)(const row16_t*)(rows16 + p0);
        row16_t row[kChunkRows];
#pragma unroll
        for (int r = 0; r < kChunkRows / 2; r++) row[r] = rows[r];

        // slide the window by 32 symbols
#pragma unroll
        for (int k = 0; k < 16; k++) W[k] = W[k + 16];
#pragma unroll
        for (int k = 0; k < 15; k++) Wodd[k] = Wodd[k + 16];
        expand(p0 + 32, 16);
        if (p0 == p_lo) {
#pragma unroll
            for (int k = 0; k < 15; k++) Wodd[k] = __builtin_amdgcn_alignbit(W[k + 1], W[k], 16);
        }
#pragma unroll
        for (int k = 15; k < 31; k++) Wodd[k] = __builtin_amdgcn_alignbit(W[k + 1], W[k], 16);
#pragma unroll
        for (int k = 0; k < 31; k++) asm volatile("" : "+v"(Wodd[k]));   // keep in VGPRs: hipcc otherwise
                                                                          // recomputes them in every odd row
#pragma unroll
        for (int r = kChunkRows / 2; r < kChunkRows; r++) row[r] = rows[r];

        uint32_t rows_with_hits = 0;   // wave-uniform
        chunk_rows(x, W, Wodd, row, sink.row_masks, rows_with_hits, lane, std::make_integer_sequence<int, kChunkRows>{});
        if (rows_with_hits) staged = drain_rows(sink, staged, rows_with_hits, (uint32_t)p0, d0 + p0, lane);
    }
    flush_hits(sink, staged, lane);
}

// ---------------------------------------------------------------------------
// after the radix sort of the keys: rewrite them as the reference's records
__global__ void ssv_keys_to_records(uint64_t* __restrict__ hits, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hits[i] = key_to_record(hits[i]);
}
__global__ void ssv_records_to_keys(uint64_t* __restrict__ hits, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hits[i] = record_to_key(hits[i]);
}

}  // namespace havac
