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
//  * TWO CELLS PER VGPR as packed int16 holding (256*score - 32768).  With that
//    scale and bias one v_pk_add_i16 with clamp does everything the FPGA's cell
//    does (device/HavacHls.cpp:370-386): saturation at -32768 IS the clamp at
//    score 0, and saturation at +32767 IS the threshold: score + match >= 256
//    exactly when the sum leaves the int16 range.  Every regular value is a
//    multiple of 256, so a crossed cell is recognisable afterwards by a non-zero
//    low byte (0x7fff), and it stays recognisable for at least one more row
//    (0x7fff + 256*m keeps the low byte unless it saturates again at the top).
//    That is why hits are looked for only once per PAIR of rows.
//  * MATCH SCORES BY v_perm_b32 straight from the reference's int8 row word: the
//    selector of a diagonal pair with symbols (a,b) is the byte pattern
//    [0x0c, a, 0x0c, b] (0x0c selects the constant 0x00), which yields
//    (M[a] << 8) | (M[b] << 24): both match scores, already multiplied by 256.
//    The 2-bit symbols are expanded into selectors once per 32-row chunk through
//    an LDS table.  The symbol window slides one position per row; even rows use
//    the aligned selector words, odd rows the words shifted by 16 bits, both
//    indexed statically in the fully unrolled chunk, so the slide costs nothing.
//  * OUTSIDE THE MATRIX (columns < 0 or >= N, rows >= nrows) the selector
//    / row yields a negative score, which pins the cell at 0 and can never hit:
//    no per-cell predicate anywhere in the hot loop.
//  * HITS are rare (about 1e-5 per cell on Dfam-like models): after each row pair
//    the 16 score registers are OR-ed and one wave-wide test looks at the low
//    bytes; only then the slow path sorts out which row crossed, repairs the
//    second row for cells that crossed on the first (they restart from 0,
//    test/softSsv/SoftSsv.cpp:43-44), and parks one mask per row in LDS.
//    Records leave through LDS staging with one atomic per burst and are put in
//    the FPGA's order afterwards.
//
// Roofline: integer VALU issue; HBM
// traffic is N/4 + 4*rows + 8*hits bytes per launch, thousands of cells per byte.
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
constexpr uint32_t kScoreZero = 0x80008000u;      // two cells at score 0: 256*0 - 32768
constexpr uint32_t kCrossedBits = 0x00010001u;    // bit 0 of either cell: set only in 0x7fff-derived values
constexpr uint32_t kPadSelector = 0x0d0c0d0cu;    // v_perm: bytes (0x00,0xff) -> 0xff00 = score -1, twice
constexpr uint32_t kPadRow = 0x80808080u;         // a row past the model: -128 for every symbol

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
// model int8 [row][A,C,G,T] -> the same words, padded with rows of -128 up to a
// whole number of chunks (a padding row can only lower a score).  The kernel
// reads the copy through the constant address space.
__global__ void ssv_pad_model(const int8_t* __restrict__ phmm, uint32_t nrows,
                              uint32_t* __restrict__ rows8, uint32_t nrows_padded) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows_padded) return;
    rows8[r] = r < nrows ? reinterpret_cast<const uint32_t*>(phmm)[r] : kPadRow;
}

// ---------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_add_pk16(uint32_t a, uint32_t b) {
    short2v r = __builtin_elementwise_add_sat(__builtin_bit_cast(short2v, a),
                                              __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}

// selector word for the two symbols in bits [shift+3 : shift] of w: bytes
// [0x0c, a, 0x0c, b] (a = low symbol) -> v_perm gives (row[a] << 8) | (row[b] << 24)
__device__ __forceinline__ uint32_t pair_selector(uint32_t w, int shift) {
    uint32_t a = (w >> shift) & 3u;
    uint32_t b = (w >> (shift + 2)) & 3u;
    return 0x000c000cu | (a << 8) | (b << 24);
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
// has bit r set for row p0 + r.  Bit q of lane l's mask is the lane's diagonal
// 4*(q & 7) + (q >> 3) (see crossed_mask), whose column on row p0 + r is
// wave_column0 + r + 32*l + that.  Returns the new number of staged records.
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
                const int diagonal = 4 * (lane & 7) + (lane >> 3);
                sink.stage[pos] = hit_key(p0 + r, (uint64_t)(wave_column0 + r + 32 * src + diagonal));
            }
            staged += __popc(m);
            if (staged > kHitStage - 32) staged = flush_hits(sink, staged, lane);
        }
    }
    return staged;
}

// The padded model is read through the constant address space: the address is
// wave-uniform, so hipcc emits s_load_dwordx16 into SGPRs (no VGPR, no vmcnt
// wait in the row loop).  Legal because the kernel never writes it.
typedef const __attribute__((address_space(4))) uint32_t* const_rows_t;

__device__ __forceinline__ uint32_t match_pair(uint32_t row, uint32_t selector) {
    return __builtin_amdgcn_perm(row, row, selector);     // (row[a] << 8) | (row[b] << 24)
}

// One bit per cell of the lane: set where the cell's low byte is non-zero (it crossed 256).
// Bit q stands for the lane's diagonal 4*(q & 7) + (q >> 3): the layout that costs least here,
// two registers per v_perm; drain_rows undoes it.
__device__ __forceinline__ uint32_t crossed_mask(const uint32_t (&v)[kRegs]) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < kRegs / 2; j++) {
        // bytes: [v[2j] low cell, v[2j] high cell, v[2j+1] low cell, v[2j+1] high cell], each 0xff or 0x00 in bit 0
        const uint32_t w = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x06040200u) & 0x01010101u;
        m |= w << j;
    }
    return m;
}

// Two model rows (R even, R+1) over the lane's 32 diagonals.  R is a template
// parameter so that every selector index is a compile-time constant.
template <int R>
__device__ __forceinline__ void row_pair_step(uint32_t (&x)[kRegs], const uint32_t (&W)[32], const uint32_t (&Wodd)[31],
                                              const uint32_t row0, const uint32_t row1, uint32_t* __restrict__ row_masks,
                                              uint32_t& rows_with_hits, int lane) {
    uint32_t y[kRegs];       // scores after row R; x becomes the scores after row R+1
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < kRegs; i++) y[i] = sat_add_pk16(x[i], match_pair(row0, W[(R >> 1) + i]));
#pragma unroll
    for (int i = 0; i < kRegs; i++) {
        x[i] = sat_add_pk16(y[i], match_pair(row1, Wodd[(R >> 1) + i]));
        any |= x[i];
    }
    if (__builtin_expect(__any((any & kCrossedBits) != 0), 0)) {
        const uint32_t mask0 = crossed_mask(y);                       // crossed 256 on row R
        const uint32_t mask1 = crossed_mask(x) & ~mask0;              // on row R+1 (a row-R mark is still on x)
        if (__any(mask0 != 0)) {
            // a cell that crossed on row R restarts from 0 (SoftSsv.cpp:43-44): redo its row R+1 from there
#pragma unroll
            for (int i = 0; i < kRegs; i++) {
                const uint32_t s0 = (y[i] & kCrossedBits) * 0xffffu;   // 0xffff over each cell marked on row R
                const uint32_t sx = (x[i] & kCrossedBits) * 0xffffu;   // ... marked on either row
                const uint32_t fresh = sat_add_pk16(kScoreZero, match_pair(row1, Wodd[(R >> 1) + i]));
                const uint32_t repl = (fresh & s0) | (kScoreZero & ~s0);
                x[i] = (repl & sx) | (x[i] & ~sx);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kRegs; i++) {
                const uint32_t sx = (x[i] & kCrossedBits) * 0xffffu;
                x[i] = (kScoreZero & sx) | (x[i] & ~sx);               // crossed cells restart at 0
            }
        }
        row_masks[R * 64 + lane] = mask0;
        row_masks[(R + 1) * 64 + lane] = mask1;
        rows_with_hits |= 3u << R;
    }
}

template <int... P>
__device__ __forceinline__ void chunk_rows(uint32_t (&x)[kRegs], const uint32_t (&W)[32], const uint32_t (&Wodd)[31],
                                           const uint32_t (&row)[kChunkRows], uint32_t* __restrict__ row_masks,
                                           uint32_t& rows_with_hits, int lane, std::integer_sequence<int, P...>) {
    (row_pair_step<2 * P>(x, W, Wodd, row[2 * P], row[2 * P + 1], row_masks, rows_with_hits, lane), ...);
}

// Selector words of the 4 symbols of one packed byte, through a 256-entry LDS
// table (2 KiB per workgroup, built once): two words per ds_read_b64 instead
// of ten VALU instructions.
__device__ __forceinline__ uint2 byte_selectors(const uint2* __restrict__ lut, uint32_t w, int byte) {
    const uint32_t off = (byte == 0 ? (w << 3) : (w >> (8 * byte - 3))) & 0x7f8u;   // byte value * 8
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(lut) + off);
}

__global__ __launch_bounds__(64 * kWavesPerBlock, 4)   // 4 waves per SIMD: at most 128 VGPRs
void ssv_diag_kernel(const uint8_t* __restrict__ seq, const int64_t nsymbols, const uint32_t* __restrict__ rows8,
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
            // positions outside [0, N) score -1: pins the cell at 0, never hits
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
        const const_rows_t rows = (const_rows_t)(rows8 + p0);
        uint32_t row[kChunkRows];
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
        chunk_rows(x, W, Wodd, row, sink.row_masks, rows_with_hits, lane, std::make_integer_sequence<int, kChunkRows / 2>{});
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
