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
//    low byte (0x7fff), and it stays recognisable for at least one more step
//    (0x7fff + 256*m keeps the low byte unless it saturates again at the top).
//    That is why hits are looked for only once per PAIR of steps.
//  * SKEWED PAIRS.  The two cells of a register sit on adjacent diagonals, and
//    the high cell runs ONE ROW BEHIND the low cell: at step t the low cell is
//    (row t, diagonal e) and the high cell (row t-1, diagonal e+1) -- the same
//    column, hence the same sequence symbol.  The match word of a register at
//    step t therefore depends on ONE symbol: (M[t][a] << 8) | (M[t-1][a] << 24).
//  * MATCH WORDS COME FROM LDS, NOT FROM THE VALU.  Two consecutive steps of a
//    register use two consecutive symbols (a, b), so a 16-entry x 8-byte table
//    per step pair, indexed by the 4-bit code of the symbol pair, returns both
//    steps' match words in one conflict-free ds_read_b64 (16 entries x 2 banks
//    = 32 distinct banks).  The wave rebuilds its 16 tables (2.2 KB) once per
//    32-step chunk with 5 v_perm_b32 per lane.  A 17th entry per table scores
//    -128 for "outside the matrix" (columns < 0 or >= N), which pins a cell at
//    0 and can never hit: the hot loop has no per-cell predicate and no separate
//    edge path.  The same entry serves the optional separator mask (boundary
//    mode): a masked symbol pair takes 256 off every diagonal that crosses it,
//    i.e. resets it.  What remains on the VALU per register and step pair is two
//    v_pk_add_i16 and half a v_or3_b32.
//  * The symbol window slides one position per step; the 32 step-pair codes a
//    lane needs per chunk are addresses (code*8 + table base) held in VGPRs and
//    indexed statically in the fully unrolled chunk, so the slide is free.
//  * HITS are rare (about 1e-5 per cell on Dfam-like models): after each step
//    pair the 16 score registers are OR-ed and one wave-wide test looks at the
//    low bytes; only then the slow path runs, one marked lane at a time and on
//    the SCALAR unit (v_readlane of the lane's registers, everything else SALU):
//    it sorts out which step crossed, puts the crossed cells back to score 0
//    (0x7fff + 1 = 0x8000; test/softSsv/SoftSsv.cpp:43-44) -- taking the second
//    step again for cells that crossed on the first -- and writes the records
//    into an LDS stage.  Records leave with one returning atomic per burst of
//    128 and one per block at the end of the tile, and are put in the FPGA's
//    order afterwards.
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
constexpr int kChunkRows = 32;                    // steps per unrolled chunk
constexpr int kChunkPairs = kChunkRows / 2;
constexpr int kWavesPerBlock = 4;
constexpr int kModelSlack = 40;                   // padding rows kept behind the padded model (table look-ahead)
constexpr uint32_t kScoreZero = 0x80008000u;      // two cells at score 0: 256*0 - 32768
constexpr uint32_t kCrossedBits = 0x00010001u;    // bit 0 of either cell: set only in 0x7fff-derived values
constexpr uint32_t kOutsideWord = 0x80008000u;    // match word "score -128" for both cells
constexpr uint32_t kPadRow = 0x80808080u;         // a row outside the model: -128 for every symbol

// per-wave LDS: 16 step-pair tables of 17 entries x 8 B, the record stage
constexpr int kPairStride = 17 * 8;               // 16 symbol-pair codes + the "outside the matrix" entry
constexpr uint32_t kOutsideCode = 16 * 8;         // byte offset of that entry
constexpr int kTableBytes = 2304;                 // 16 x 136 = 2176, rounded up to a multiple of 128
constexpr int kHitStage = 128;                    // records staged per wave (1 KiB)

// sort key of a hit: (segment, row, column-in-segment) -- the FPGA's emission
// order (device/HavacHls.cpp:151-152,264; device/HitReporting.cpp:178-337)
// The row field is only as wide as the model needs (row_bits), so that the radix sort has no dead passes.
__device__ __forceinline__ uint64_t hit_key(uint32_t row, uint64_t column, uint32_t row_bits) {
    // 12288 = 3 * 4096 and columns stay below 2^34 (packed sequence < 4 GiB): a 32-bit division by 3
    const uint32_t seg = (uint32_t)(column >> 12) / 3u;
    const uint32_t in_seg = (uint32_t)column - seg * 12288u;
    return ((uint64_t)seg << (14 + row_bits)) | ((uint64_t)row << 14) | in_seg;
}

// key -> the reference's packed record (device/HitReporting.cpp:421-430)
__device__ __forceinline__ uint64_t key_to_record(uint64_t key, uint32_t row_bits) {
    uint64_t in_seg = key & 0x3fffull;
    uint64_t row = (key >> 14) & ((1ull << row_bits) - 1ull);
    uint64_t seg = key >> (14 + row_bits);
    return in_seg | (seg << 14) | (row << 40);
}
__device__ __forceinline__ uint64_t record_to_key(uint64_t rec, uint32_t row_bits) {
    uint64_t in_seg = rec & 0x3fffull;
    uint64_t seg = (rec >> 14) & 0x3ffffffull;
    uint64_t row = rec >> 40;
    return (seg << (14 + row_bits)) | (row << 14) | in_seg;
}

// ---------------------------------------------------------------------------
// model int8 [row][A,C,G,T] -> the same words shifted by one: out[0] is "row -1"
// (all -128), out[1 + r] is row r, and everything from row nrows on is -128
// again (a padding row can only lower a score).  nrows_padded + kModelSlack words.
__global__ void ssv_pad_model(const int8_t* __restrict__ phmm, uint32_t nrows,
                              uint32_t* __restrict__ rows, uint32_t nwords) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    rows[i] = (i >= 1 && i <= nrows) ? reinterpret_cast<const uint32_t*>(phmm)[i - 1] : kPadRow;
}

// ---------------------------------------------------------------------------
// Text -> 2-bit symbols.  One thread packs 16 characters into one 32-bit word with
// the layout of host/sequence/SequencePreprocessor.cpp:46-57 (symbol i in byte i/4, bits (i%4)*2).  For a/c/g/t in
// either case the code is a function of bits 1..2 of the character: A 0x41, C 0x43, G 0x47, T 0x54 give
// (c >> 1) & 3 = 0, 1, 3, 2, and x ^ (x >> 1) turns that into 0, 1, 2, 3.  Other characters produce some value
// here and are overwritten by ssv_patch_symbols.  `chars` starts at column `first_column` (a multiple of 16);
// columns at or beyond `nchars` are symbol 0 (the reference's zero padding, :41).
__global__ void ssv_pack_chars(const uint8_t* __restrict__ chars, uint64_t first_column, uint64_t nchars,
                               uint64_t ncolumns, uint32_t* __restrict__ packed) {
    const uint64_t word = first_column / 16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t column = word * 16;
    if (column >= ncolumns) return;
    uint32_t out = 0;
    if (column + 16 <= nchars) {
        const uint4 v = *reinterpret_cast<const uint4*>(chars + (column - first_column));
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t t = (w[k] >> 1) & 0x03030303u;          // per character: 0, 1, 3, 2 for A, C, G, T
            t ^= (t >> 1) & 0x01010101u;                      // -> 0, 1, 2, 3
            const uint32_t byte = (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xffu;
            out |= byte << (8 * k);
        }
    } else {
        for (int i = 0; i < 16; i++) {
            if (column + i >= nchars) break;
            uint32_t t = (chars[column - first_column + i] >> 1) & 3u;
            t ^= t >> 1;
            out |= t << (2 * i);
        }
    }
    packed[word] = out;
}

// columns that are not a/c/g/t: the symbol the host drew for them
__global__ void ssv_patch_symbols(const uint64_t* __restrict__ columns, const uint8_t* __restrict__ symbols,
                                  uint64_t npatches, uint32_t* __restrict__ packed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npatches) return;
    const uint64_t column = columns[i];
    const uint32_t shift = (uint32_t)(column % 16) * 2;
    atomicAnd(&packed[column / 16], ~(3u << shift));
    atomicOr(&packed[column / 16], (uint32_t)(symbols[i] & 3u) << shift);
}

// ---------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_words_t;
typedef __attribute__((address_space(3))) u32x2* lds_words_out_t;

__device__ __forceinline__ uint32_t sat_add_pk16(uint32_t a, uint32_t b) {
    short2v r = __builtin_elementwise_add_sat(__builtin_bit_cast(short2v, a),
                                              __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}

// 8 bytes = 32 symbols at symbol offset `pos` (multiple of 32); zero outside the buffer
__device__ __forceinline__ uint2 load_symbols(const uint8_t* __restrict__ seq, int64_t nsymbols, int64_t pos,
                                              bool checked) {
    if (checked && (pos < 0 || pos + 32 > nsymbols)) return make_uint2(0u, 0u);
    return *reinterpret_cast<const uint2*>(seq + (pos >> 2));
}

// ---- per-wave LDS ------------------------------------------------------------
struct __attribute__((aligned(128))) WaveLds {
    uint8_t table[kTableBytes];            // match words of the current chunk
    uint64_t stage[kHitStage];             // records waiting for the next burst
};

// ---- hit queue --------------------------------------------------------------
// Counterpart of the FPGA's five-stage hit sieve (device/HitReporting.cpp:12-417).
// A wave that sees a crossing turns it into a record at once (step_pair's slow
// path, on the scalar unit) and stages it in LDS; staged records are
// appended to the global queue in bursts: one returning atomic per burst
// instead of one per hit (a single counter word sustains only ~90 returning
// atomics per microsecond chip-wide, which capped the first version of this
// kernel at ~90 M hits/s).
struct HitSink {
    uint64_t* hits;                // global queue of sort keys (hit_key)
    unsigned long long* hit_count; // records found so far (may run past capacity)
    uint64_t hit_capacity;
    WaveLds* lds;
    int64_t col_begin, col_end;    // only hits in these columns are reported (the shard's own columns)
    uint32_t row_bits;             // width of the row field of the sort key
};

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

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
        if (idx < sink.hit_capacity) sink.hits[idx] = sink.lds->stage[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return 0;
}

// A full stage leaves with one returning atomic (a real call: rare, and its registers stay out of the hot loop).
__device__ __noinline__ uint32_t flush_full(const HitSink sink, uint32_t staged, int lane) {
    return flush_hits(sink, __builtin_amdgcn_readfirstlane(staged), lane);
}

// Two int16 saturating adds on the scalar unit (the slow path's copy of v_pk_add_i16 ... clamp).
__device__ __forceinline__ uint32_t scalar_sat_add_pk16(uint32_t a, uint32_t b) {
    int lo = (int)(short)(a & 0xffffu) + (int)(short)(b & 0xffffu);
    int hi = (int)(short)(a >> 16) + (int)(short)(b >> 16);
    lo = lo < -32768 ? -32768 : (lo > 32767 ? 32767 : lo);
    hi = hi < -32768 ? -32768 : (hi > 32767 ? 32767 : hi);
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}

// Records of one lane's crossings at step t.  Bit i of `marks` = the low cell of register i, bit 16 + i = its high
// cell.  At step t the low cell of register i of lane l is (row t, column c) and the high cell (row t - 1, column c),
// with c = wave_diag0 + 32*l + 2*i + t.  Everything here is wave-uniform: it runs on the scalar unit, which this
// VALU-bound kernel leaves idle; per record the VALU only moves the key to the LDS stage.
__device__ __forceinline__ uint32_t emit_marks(const HitSink& sink, uint32_t staged, uint32_t marks, uint32_t t,
                                               uint32_t l, int64_t wave_diag0, int lane) {
    const auto lds = (__attribute__((address_space(3))) WaveLds*)sink.lds;
    while (marks) {
        const uint32_t q = (uint32_t)__builtin_ctz(marks);
        marks &= marks - 1;
        const uint32_t reg = q & 15, high = q >> 4;
        const int64_t column = wave_diag0 + (int64_t)(32 * l + 2 * reg + t);
        if (column < sink.col_begin || column >= sink.col_end) continue;   // halo columns belong to the neighbouring shard
        const uint64_t key = hit_key(t - high, (uint64_t)column, sink.row_bits);
        if (lane == 0) lds->stage[staged] = key;
        staged++;
        if (staged == kHitStage) staged = flush_full(sink, staged, lane);
    }
    return staged;
}

// both steps' match words of one register: one ds_read_b64, table of step pair P, entry `code_addr`
template <int P>
__device__ __forceinline__ u32x2 match_words(uint32_t code_addr) {
    return *(lds_words_t)(uintptr_t)(code_addr + P * kPairStride);
}

// Steps 2P and 2P+1 of the chunk over the lane's 32 diagonals.  P is a template parameter so that
// every window index and LDS offset is a compile-time constant.
//
// The scores ping-pong between two register sets: on entry `cur` holds the scores, the first step
// updates `cur` in place (it then holds the scores after step 2P, which the slow path needs), the
// second step writes `nxt`; the next pair is instantiated with the sets swapped.  The first add is
// written as read-modify-write asm, which pins a score to its VGPR through the unrolled chunk (left
// to itself hipcc re-homes the 16 registers with v_mov after every pair; measured on C2: both adds
// plain 2.62 ms, both asm 2.49 ms, first asm + second plain 2.39 ms).
template <int P>
__device__ __forceinline__ void step_pair(uint32_t (&cur)[kRegs], uint32_t (&nxt)[kRegs], const uint32_t (&C)[32],
                                          const HitSink& sink, uint32_t& staged, uint32_t step0, int64_t wave_diag0,
                                          int lane) {
    uint32_t any = 0;
    u32x2 m[kRegs];
#pragma unroll
    for (int i = 0; i < kRegs; i++) m[i] = match_words<P>(C[P + i]);
    __builtin_amdgcn_sched_barrier(0);      // all 16 reads in flight before the first add waits (hipcc otherwise staggers them)
#pragma unroll
    for (int i = 0; i < kRegs; i++) asm("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(cur[i]) : "v"(m[i].x));
#pragma unroll
    for (int i = 0; i < kRegs; i++) {
        nxt[i] = sat_add_pk16(cur[i], m[i].y);
        any |= nxt[i];
    }
    if (__builtin_expect(__any((any & kCrossedBits) != 0), 0)) {
        // (the lane mask is taken again inside: the hot test stays v_cmp -> vcc -> s_cbranch_vccnz)
        unsigned long long lanes = __ballot((any & kCrossedBits) != 0);
        // Slow path, one lane at a time on the scalar unit (typically one lane, one register).  A crossed cell holds
        // exactly 0x7fff; "restart from 0" (SoftSsv.cpp:43-44) is + 1 on that int16 (0x8000 = score 0).  A cell that
        // crossed on the first step still shows its mark on `nxt`; it gets the + 1 on `cur` and takes the second step again.
        do {
            const uint32_t l = (uint32_t)__builtin_ctzll(lanes);
            lanes &= lanes - 1;
            const bool mine = (uint32_t)lane == l;
            uint32_t first = 0, second = 0;     // bit i: low cell of register i, bit 16 + i: its high cell
#pragma unroll
            for (int i = 0; i < kRegs; i++) {
                uint32_t n = __builtin_amdgcn_readlane(nxt[i], l);
                if (n & kCrossedBits) {
                    uint32_t c = __builtin_amdgcn_readlane(cur[i], l);
                    const uint32_t f = c & kCrossedBits;
                    if (f) {
                        c += f;
                        n = scalar_sat_add_pk16(c, __builtin_amdgcn_readlane(match_words<P>(C[P + i]).y, l));
                        first |= f << i;
                    }
                    const uint32_t sec = n & kCrossedBits;
                    n += sec;
                    second |= sec << i;
                    nxt[i] = mine ? n : nxt[i];
                }
            }
            staged = emit_marks(sink, staged, first, step0 + 2 * P, l, wave_diag0, lane);
            staged = emit_marks(sink, staged, second, step0 + 2 * P + 1, l, wave_diag0, lane);
        } while (lanes);
    }
}

// pairs First .. First+N-1; the scores are in `a` on entry and, N being even, in `a` again on exit
template <int First, int... I>
__device__ __forceinline__ void step_pairs(uint32_t (&a)[kRegs], uint32_t (&b)[kRegs], const uint32_t (&C)[32],
                                           const HitSink& sink, uint32_t& staged, uint32_t step0, int64_t wave_diag0,
                                           int lane, std::integer_sequence<int, I...>) {
    static_assert(sizeof...(I) % 2 == 0, "an even number of pairs returns the scores to the first set");
    ((I % 2 == 0 ? step_pair<First + I>(a, b, C, sink, staged, step0, wave_diag0, lane)
                 : step_pair<First + I>(b, a, C, sink, staged, step0, wave_diag0, lane)), ...);
}

// selector of the match word of symbol a: bytes [0x0c, a, 0x0c, 4 + a]; with v_perm(S0 = row t-1, S1 = row t) it
// yields (row_t[a] << 8) | (row_tm1[a] << 24)
__device__ __forceinline__ uint32_t word_selector(uint32_t a) { return 0x040c000cu + a * 0x01000100u; }

// 5 waves per SIMD: 96 VGPRs, no scratch.  Measured on C2 with the scalar slow path: 4 waves (106 VGPRs) 2.21 ms,
// 5 waves 2.12 ms; without hits 2.02 / 1.98 ms.  (With the earlier vector slow path the kernel needed 121 VGPRs and
// 5 waves spilled badly; 3 waves are 7 % slower.)
__global__ __launch_bounds__(64 * kWavesPerBlock, 5)
void ssv_diag_kernel(const uint8_t* __restrict__ seq, const int64_t nsymbols, const uint32_t* __restrict__ rows,
                     const uint32_t nrows_padded, const int64_t first_diag, const uint32_t tile_begin,
                     const uint32_t tile_end, const int64_t col_begin, const int64_t col_end,
                     uint64_t* __restrict__ hits, unsigned long long* __restrict__ hit_count,
                     const uint64_t hit_capacity, const uint32_t* __restrict__ abort_flag,
                     const uint16_t* __restrict__ pair_mask, const uint32_t row_bits) {
    __shared__ WaveLds wave_lds[kWavesPerBlock];

    const int lane = threadIdx.x & 63;
    // readfirstlane: everything derived from the tile index is wave-uniform (SALU branches, scalar address math)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds* const lds = &wave_lds[wave];
    const HitSink sink{hits, hit_count, hit_capacity, lds, col_begin, col_end, row_bits};
    uint32_t staged = 0;          // wave-uniform

    // LDS byte address of this wave's tables (a multiple of 128: code*8 is OR-ed into it)
    const uint32_t table_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds->table;
    // this lane's share of the table build: the four entries (a, b) with b = lane & 3 of step pair lane >> 2
    const uint32_t my_pair = lane >> 2, my_b = lane & 3;
    const uint32_t sel_second = word_selector(my_b);
    const uint32_t my_entries_addr = table_base + my_pair * kPairStride + my_b * 32;
    if (my_b == 0)        // the "outside the matrix" entry of each table never changes
        *(lds_words_out_t)(uintptr_t)(table_base + my_pair * kPairStride + kOutsideCode) = u32x2{kOutsideWord, kOutsideWord};

    const uint32_t tile = tile_begin + blockIdx.x * kWavesPerBlock + wave;
    const int64_t d0 = first_diag + (int64_t)tile * kTileDiags;     // wave's first diagonal
    const int64_t dl = d0 + lane * kDiagsPerLane;                   // lane's first diagonal
    // steps whose cells of this tile can lie inside the matrix
    int64_t p_lo = -d0 - kTileDiags;                                // first chunk touching column >= 0
    if (p_lo < 0) p_lo = 0;
    int64_t p_hi = col_end - d0;                                    // first chunk entirely right of the shard's columns
    if (p_hi > (int64_t)nrows_padded) p_hi = nrows_padded;
    // no early return: every wave of the block meets the others at the final flush
    if (tile < tile_end && p_lo < p_hi) {
        uint32_t x[kRegs], x2[kRegs];     // the scores and their ping-pong partner (see step_pair)
#pragma unroll
        for (int i = 0; i < kRegs; i++) x[i] = x2[i] = kScoreZero;

        // Window of the current chunk, symbol positions [j, j+64) with j = dl + p0: C[k] is the LDS address of the
        // table entry of the symbol pair (j+2k, j+2k+1).  The upper half of one chunk's window is the lower half of the
        // next, so each chunk expands only the 32 new symbols.
        uint32_t C[32];
        // packed symbols [dl+rel, dl+rel+32) in x,y; in z the 16 separator bits of those symbol pairs (boundary mode);
        // rel is wave-uniform
        auto fetch_symbols = [&](int64_t rel) -> uint3 {
            const bool edge = (d0 + rel < 0) || (d0 + rel + kTileDiags > nsymbols);   // the wave's 2048 positions
            const int64_t pos = dl + rel;
            const uint2 w = load_symbols(seq, nsymbols, pos, edge);
            uint32_t separators = 0;
            if (pair_mask && pos >= 0 && pos + 32 <= nsymbols) separators = pair_mask[pos >> 5];
            return make_uint3(w.x, w.y, separators);
        };
        auto expand = [&](const uint3 packed, int64_t rel) {  // -> C[16 .. 32)
            const bool edge = (d0 + rel < 0) || (d0 + rel + kTileDiags > nsymbols);
            const int64_t pos = dl + rel;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t w = k < 8 ? packed.x : packed.y;
                const int sh = (k & 7) * 4;                                  // the pair's 4 bits start here
                const uint32_t code8 = sh == 0 ? (w << 3) : (w >> (sh - 3)); // code * 8 in bits [6:3]
                C[16 + k] = (code8 & 0x78u) | table_base;
            }
            if (edge) {
                // positions outside [0, N) use the entry that scores -1: pins the cell at 0, never hits
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int64_t q = pos + 2 * k;     // q and q+1 are in or out together (pos, N even)
                    if (q < 0 || q >= nsymbols) C[16 + k] = table_base + kOutsideCode;
                }
            }
            if (pair_mask && __any(packed.z != 0)) {
                // boundary mode: a separator pair scores -128 twice on every diagonal through it = a reset
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if ((packed.z >> k) & 1u) C[16 + k] = table_base + kOutsideCode;
            }
        };
        // A chunk's 16 step-pair tables: entry (a,b) of pair P = { word(step 2P, a), word(step 2P+1, b) } with
        // word(t, a) = (M[t][a] << 8) | (M[t-1][a] << 24); rows[] is shifted by one, so rows[t] is M[t-1].
        // Four lanes share a step pair: they fetch its three model rows and each writes the entries of one b.
        struct ModelRows { uint32_t r0, r1, r2; };
        auto fetch_rows = [&](int64_t p0) -> ModelRows {
            const uint32_t* r = rows + p0 + 2 * my_pair;
            return ModelRows{r[0], r[1], r[2]};
        };
        auto build_tables = [&](const ModelRows r) {
            const uint32_t second = __builtin_amdgcn_perm(r.r1, r.r2, sel_second);
            const lds_words_out_t out = (lds_words_out_t)(uintptr_t)my_entries_addr;
#pragma unroll
            for (int a = 0; a < 4; a++) out[a] = u32x2{__builtin_amdgcn_perm(r.r0, r.r1, word_selector(a)), second};
        };
        expand(fetch_symbols(p_lo), p_lo);
        // The global loads of a chunk (12 B of model rows and 8 B of symbols per lane) are issued at the END of the chunk
        // before it and consumed at its top: no register carries them across the 16 step pairs (held there they cost six
        // VGPRs, i.e. scratch at 96), and the latency that is exposed this way is covered by the other four waves of
        // the SIMD (measured: 2.12 ms against 2.16 ms with the loads issued a whole chunk ahead).
        ModelRows next_rows = fetch_rows(p_lo);
        uint3 next_symbols = fetch_symbols(p_lo + 32);

        for (int64_t p0 = p_lo; p0 < p_hi; p0 += kChunkRows) {
            // abort: a device word, read past the caches every 2048 rows (never on a wave's first chunk, so
            // short models pay nothing)
            if (abort_flag && ((p0 & 2047) == 0) && p0 != p_lo &&
                __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
            build_tables(next_rows);
            // slide the window by 32 symbols
#pragma unroll
            for (int k = 0; k < 16; k++) C[k] = C[k + 16];
            expand(next_symbols, p0 + 32);

            step_pairs<0>(x, x2, C, sink, staged, (uint32_t)p0, d0, lane, std::make_integer_sequence<int, kChunkPairs>{});
            asm volatile("" ::: "memory");                        // keeps hipcc from hoisting the loads above the step pairs
            next_rows = fetch_rows(p0 + kChunkRows);              // rows[] has kModelSlack words behind the model
            next_symbols = fetch_symbols(p0 + kChunkRows + 32);
        }
        if (p_hi == (int64_t)nrows_padded) {
            // the high cells run one row behind: one more step gives them the model's last row
            build_tables(next_rows);                               // fetched for p_hi by the last chunk
#pragma unroll
            for (int k = 0; k < 16; k++) C[k] = C[k + 16];
            step_pair<0>(x, x2, C, sink, staged, (uint32_t)p_hi, d0, lane);
        }
    }   // this wave's tile

    // What is still staged goes out with ONE returning atomic per block, not per wave: the single counter word
    // sustains ~90 returning atomics per microsecond chip-wide, and with one per wave every model shorter than ~200
    // rows was bound by that rate, not by the VALU (48,836 tiles with a hit each = 0.55 ms on C2's 100 Mbp).
    __shared__ uint32_t block_staged[kWavesPerBlock];
    __shared__ unsigned long long block_base;
    if (lane == 0) block_staged[wave] = staged;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kWavesPerBlock; w++) total += block_staged[w];
        block_base = total ? atomicAdd(hit_count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = block_base;
    for (uint32_t w = 0; w < wave; w++) base += block_staged[w];
    for (uint32_t i = lane; i < staged; i += 64)
        if (base + i < hit_capacity) hits[base + i] = lds->stage[i];
}

// ---------------------------------------------------------------------------
// after the radix sort of the keys: rewrite them as the reference's records
__global__ void ssv_keys_to_records(uint64_t* __restrict__ hits, uint64_t n, uint32_t row_bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hits[i] = key_to_record(hits[i], row_bits);
}
__global__ void ssv_records_to_keys(uint64_t* __restrict__ hits, uint64_t n, uint32_t row_bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hits[i] = record_to_key(hits[i], row_bits);
}

}  // namespace havac
