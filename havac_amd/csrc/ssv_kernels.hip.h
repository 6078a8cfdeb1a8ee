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
//    low byte (0x7fff), and it stays recognisable until it saturates at the
//    bottom, i.e. until the scores added after the crossing sum to -256 or less.
//    That is why hits are looked for only once per FOUR steps wherever the three
//    model rows that can follow a crossing cannot sum below -255 (ssv_prepare_model
//    marks those four-step windows, a byte per 32-row chunk; elsewhere, and where a
//    separator pair or the matrix's edge lies in the wave's window under a
//    separator mask, every two steps).
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
//    32-step chunk with 5 v_perm_b32 per lane.  A 17th entry per table is for
//    "outside the matrix" (columns < 0 or >= N): it scores 0, so a cell there
//    keeps what it has and can never hit, and the hot loop has no per-cell
//    predicate and no separate edge path.  With a separator mask (boundary
//    mode) the same entry scores -128: a masked symbol pair takes 256 off every
//    diagonal that crosses it, i.e. resets it.  What remains on the VALU per
//    register and four steps is four v_pk_add_i16 and half a v_or3_b32.
//  * The symbol window slides one position per step; the table addresses a lane
//    needs (code*8 + table base, one per symbol pair) are held in VGPRs and
//    indexed statically in the fully unrolled chunk, so the slide is free.  A
//    window of four steps uses 17 consecutive addresses; the upper half of the
//    chunk's 32 is expanded two entries per window (one SDWA v_or each), so
//    only ~19 are alive at a time.
//  * HITS are rare (about 1e-5 per cell on Dfam-like models): after each window
//    the 16 score registers are OR-ed and one wave-wide test looks at the low
//    bytes; only then the slow path runs, one marked lane and register at a
//    time (found by halving) and on the SCALAR unit: the register's four steps are taken again
//    from the scores the window started with (kept in the other register set),
//    crossings are reported and put back to score 0 on the way (0x7fff + 1 =
//    0x8000; test/softSsv/SoftSsv.cpp:43-44), and the exact result is written
//    over the lane's register.  Records go to an LDS stage and leave with one
//    returning atomic per burst of 128; what a block still has staged when it
//    ends leaves with one atomic per block, or -- short models, whose blocks end
//    faster than one counter word can answer -- through the block's own slots of
//    a side buffer and a small gather kernel (block tails).  The records are put
//    in the FPGA's order afterwards.
//  * SIX WAVES PER SIMD: 80 VGPRs (two score sets of 16, 16 match words in
//    flight, ~19 window addresses, a few temporaries) and nothing else kept in
//    a register through the hot loop: no lane number (per-lane constants come
//    from LDS with ds_read_addtid_b32), no launch parameter that is needed
//    rarely (SsvRare: read from the kernarg segment where it is used), no
//    scratch.
//
//  * WHO TAKES WHICH TILE: workgroup i serves partition i mod 8 -- workgroups are dealt round-robin over the eight XCDs,
//    so a partition (a run of adjacent tiles, of equal work) lives in one L2: adjacent tiles stream the same bytes of
//    the sequence 64 chunks apart.  Whole tiles first; where tiles are tall, only the tiles that run last are cut by
//    rows (tapered cuts, 2 KB handed from wave to wave per cut), or -- fewer tiles than wave slots -- every tile into
//    uniform blocks.  See "work distribution" and "items" below; the plan is made on the host (havac_dev.hip, plan_launch).
//
// Roofline: integer VALU issue and, at the same ceiling, LDS bandwidth (SURVEY.md
// section 8d); HBM traffic is N/4 + 4*rows + 8*hits bytes
// per launch, thousands of cells per byte.
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
#ifndef HAVAC_BATCH
#define HAVAC_BATCH 8
#endif
#ifndef HAVAC_WAVES_PER_SIMD
#define HAVAC_WAVES_PER_SIMD 6
#endif
#ifndef HAVAC_WAVES_PER_BLOCK
#define HAVAC_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = HAVAC_WAVES_PER_BLOCK;
constexpr int kModelSlack = 40;                   // padding rows kept behind the padded model (table look-ahead)
constexpr uint32_t kScoreZero = 0x80008000u;      // two cells at score 0: 256*0 - 32768
constexpr uint32_t kCrossedBits = 0x00010001u;    // bit 0 of either cell: set only in 0x7fff-derived values
constexpr uint32_t kOutsideReset = 0x80008000u;   // match word "score -128" for both cells (separator pairs: two of them reset a diagonal)
constexpr uint32_t kOutsideNeutral = 0u;          // match word "score 0": columns outside the matrix when there are no separators
constexpr uint32_t kPadRow = 0u;                  // a row outside the model scores 0 for every symbol: it changes nothing and can never hit
constexpr int kMaxRowCuts = 32;                   // row blocks of individual height in front of the uniform ones (SsvRare::row_cut)
constexpr int kMaxWalkRounds = 8;                 // rounds of runs of a resident-table launch (SsvRare::walk_len)
constexpr int kWindowSteps = 4;                   // steps between two hit tests where the model allows it (ssv_prepare_model)

// per-wave LDS: 16 step-pair tables of 17 entries x 8 B, the record stage
constexpr int kPairStride = 17 * 8;               // 16 symbol-pair codes + the "outside the matrix" entry
constexpr uint32_t kOutsideCode = 16 * 8;         // byte offset of that entry
constexpr int kTableBytes = 2304;                 // 16 x 136 = 2176, rounded up to a multiple of 128
constexpr int kHitStage = 128;                    // records staged per wave (1 KiB)
constexpr int kTailSlots = 128;                   // records a block may leave in its slots of the side buffer when it ends (block tails)
constexpr int kTicketStride = 16;                 // the ticket counters and the fault word sit in cache lines of their own
constexpr int kTicketCounters = 8;                // one per partition of a launch (SsvRare::parts_log2 <= 3)

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
// (all 0), out[1 + r] is row r, and everything from row nrows on is 0 again (a
// padding row leaves every score as it is and can never hit; it must not LOWER a
// score either, or a crossing on the model's last rows could lose its mark before
// the next hit test, see step_window).  nrows_padded + kModelSlack words.
__device__ __forceinline__ uint32_t padded_model_row(const uint32_t* __restrict__ phmm_rows, uint32_t nrows, uint32_t i) {
    return (i >= 1 && i <= nrows) ? phmm_rows[i - 1] : kPadRow;
}

// Which four-step windows may test for hits at their end only instead of in the middle too.  A crossed cell is recognised by its
// low byte (0x7fff), and it loses that mark only by saturating at the bottom, i.e. when the scores added after the
// crossing sum to -256 or less.  Inside a window of four steps at most three steps follow a crossing, so a window is
// safe when the three rows that can follow a crossing in it (the high cells run one row behind) cannot sum below -255 whatever the
// symbols are.  One bit per window = one byte per 32-row chunk (round 5; before: one bit per chunk -- a model with one strongly
// negative stretch per chunk tested every two steps everywhere, +8.5 % on C2); the same kernel writes the padded model.
// ONE launch prepares a pass: thread i writes word i of the padded model, thread c the flags of chunk c (from the model
// itself, not from the padded copy: the two halves do not depend on each other), thread 0 clears the hit counter.  As
// three dispatches (memset, pad, flags) in front of every SSV kernel the preparation cost ~30 us of dispatch gaps per
// pass -- 1.5 % of a C2 step -- for 10 us of work.
// Round 5: it also clears everything the pass's kernels count in -- the hit counter, the tickets of cut tiles, the ordering's
// shared words (`control`, ncontrol words) and the hand-off counts of cut tiles (`handoff`, nhandoff words): no fill is enqueued
// around the launches of a pass.
__global__ void ssv_prepare_model(const int8_t* __restrict__ phmm, uint32_t nrows, uint32_t* __restrict__ rows, uint32_t nwords,
                                  uint32_t nrows_padded, uint32_t* __restrict__ flags, uint32_t nflagwords /* 0: no flags */,
                                  unsigned long long* __restrict__ hit_count, uint32_t* __restrict__ control, uint32_t ncontrol,
                                  uint32_t* __restrict__ handoff, uint32_t nhandoff) {
    const uint32_t* const phmm_rows = reinterpret_cast<const uint32_t*>(phmm);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *hit_count = 0ull;
    if (i < ncontrol) control[i] = 0u;
    for (uint32_t k = i; k < nhandoff; k += gridDim.x * blockDim.x) handoff[k] = 0u;
    if (i < nwords) rows[i] = padded_model_row(phmm_rows, nrows, i);
    if (nflagwords == 0) return;
    // one thread per chunk: a byte, bit Q = window Q (steps 4Q .. 4Q+3 of the chunk) may look for hits at its end only
    const uint32_t chunk = i;
    const uint32_t p0 = chunk * kChunkRows;
    if (chunk / 4 >= nflagwords) return;
    uint32_t bits = 0;
    if (p0 < nrows_padded + kChunkRows) {
        // model rows p0 .. p0+34 are padded rows p0+1 .. p0+35: the lowest score of each (at most 0; 0 beyond the model) ...
        int lowest[kChunkRows + 3];
#pragma unroll
        for (int k = 0; k < kChunkRows + 3; k++) {
            const uint32_t r = padded_model_row(phmm_rows, nrows, p0 + 1 + k);
            int m = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) m = min(m, (int)(int8_t)(r >> (8 * a)));
            lowest[k] = m;
        }
        // ... and the two runs of three that can follow a crossing inside window Q: a high cell that crosses at the window's first
        // step (on row 4Q - 1) adds rows 4Q, 4Q+1, 4Q+2 before the test, a low cell (on row 4Q) rows 4Q+1, 4Q+2, 4Q+3; a crossing at
        // a later step adds a suffix of one of them, which sums to no less (no `lowest` is positive)
#pragma unroll
        for (int q = 0; q < kChunkRows / kWindowSteps; q++) {
            const int high = lowest[4 * q] + lowest[4 * q + 1] + lowest[4 * q + 2], low = lowest[4 * q + 1] + lowest[4 * q + 2] + lowest[4 * q + 3];
            if (high >= -255 && low >= -255) bits |= 1u << q;
        }
    }
    reinterpret_cast<uint8_t*>(flags)[chunk] = (uint8_t)bits;
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
// Boundary-mode layout on the GPU (SURVEY.md section 8 rows f2 + f4; host counterpart: SequencePreprocessor(fasta, true)).
// Record j keeps its own columns [out_start[j], out_start[j] + len[j]) -- residues and terminator column, from
// chars[src_begin[j] ..] -- followed, at the next even column, by one separator pair; a/c/g are 0/1/2, every other
// character is T (host/test/Ssv.cpp:29-34), no rand().  One thread makes one 32-bit word (16 columns) and the byte of
// the separator bitmap that belongs to it (one bit per aligned symbol pair); columns outside every record are symbol 0,
// and every pair from `layout_end` on (the padding behind the last record) is masked.
__global__ void ssv_pack_records(const uint8_t* __restrict__ chars, const uint64_t* __restrict__ out_start,
                                 const uint64_t* __restrict__ src_begin, const uint64_t* __restrict__ len, uint32_t nrecords,
                                 uint64_t layout_end, uint64_t ncolumns, uint32_t* __restrict__ packed, uint8_t* __restrict__ mask) {
    const uint64_t word = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t col0 = word * 16;
    if (col0 >= ncolumns) return;
    // the last record that starts at or before col0 (none: the word lies before the first record, which starts at 0)
    uint32_t lo = 0, hi = nrecords;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (out_start[mid] <= col0) lo = mid + 1; else hi = mid;
    }
    uint32_t j = lo ? lo - 1 : 0;
    uint32_t out = 0, bits = 0;
    for (int i = 0; i < 16; i++) {
        const uint64_t col = col0 + i;
        while (j + 1 < nrecords && out_start[j + 1] <= col) j++;
        if (nrecords && col >= out_start[j]) {
            const uint64_t off = col - out_start[j];
            if (off < len[j]) {
                const uint32_t ch = chars[src_begin[j] + off] | 0x20u;
                const uint32_t code = ch == 'a' ? 0u : ch == 'c' ? 1u : ch == 'g' ? 2u : 3u;
                out |= code << (2 * i);
            }
            const uint64_t behind = out_start[j] + len[j];
            if ((i & 1) == 0 && col == behind + (behind & 1)) bits |= 1u << (i / 2);      // the record's separator pair
        }
        if ((i & 1) == 0 && col >= layout_end) bits |= 1u << (i / 2);
    }
    packed[word] = out;
    mask[word] = (uint8_t)bits;
}

// Second strand on the GPU (row f3; host counterpart: SequencePreprocessor::appendReverseStrand): columns [nf, 2 nf)
// repeat [0, nf) with every record's residues reverse-complemented in place (3 - symbol, order reversed); terminator,
// separator and padding columns are copied.  starts[] ascending.  Reads the first half, writes the second.
__global__ void ssv_reverse_strand(uint32_t* __restrict__ packed, uint64_t nf, const uint64_t* __restrict__ starts,
                                   const uint64_t* __restrict__ residues, uint32_t nrecords) {
    const uint64_t word = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t col0 = word * 16;
    if (col0 >= nf) return;
    auto symbol = [&](uint64_t col) -> uint32_t { return (packed[col / 16] >> (2 * (col % 16))) & 3u; };
    uint32_t lo = 0, hi = nrecords;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (starts[mid] <= col0) lo = mid + 1; else hi = mid;
    }
    uint32_t j = lo ? lo - 1 : 0;
    uint32_t out = 0;
    for (int i = 0; i < 16; i++) {
        const uint64_t col = col0 + i;
        while (j + 1 < nrecords && starts[j + 1] <= col) j++;
        uint32_t code = symbol(col);
        if (nrecords && col >= starts[j] && col - starts[j] < residues[j])
            code = 3u - symbol(starts[j] + residues[j] - 1 - (col - starts[j]));
        out |= code << (2 * i);
    }
    packed[(nf + col0) / 16] = out;
}

// ---------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_words_t;
typedef __attribute__((address_space(3))) u32x2* lds_words_out_t;

// ---- per-wave LDS ------------------------------------------------------------
struct __attribute__((aligned(128))) WaveLds {
    uint8_t table[kTableBytes];            // match words of the current chunk
    uint64_t stage[kHitStage];             // records waiting for the next burst
};

// ---- hit queue --------------------------------------------------------------
// Counterpart of the FPGA's five-stage hit sieve (device/HitReporting.cpp:12-417).
// A wave that sees a crossing turns it into a record at once (window_slow, on the
// scalar unit) and stages it in LDS; staged records are appended to the global
// queue in bursts: one returning atomic per burst instead of one per hit (a single
// counter word sustains only ~90 returning atomics per microsecond chip-wide, which
// capped the first version of this kernel at ~90 M hits/s).
// What the kernel needs RARELY -- when a hit is reported, a stage is flushed, a row block handed over, an abort polled --
// is not kept in SGPRs through the hot loop (with it there the kernel needed ~130 SGPRs: hipcc parked 30 of them in the
// lanes of a VGPR and fetched them back with ~10 v_readlane per chunk).  It is the kernel's FIRST argument, never named
// in the kernel, and read from the kernarg segment (scalar loads, constant cache) where it is used.
// ---- per-cell trace ----------------------------------------------------------------------------------------------------
// Counterpart of the reference's HAVAC_PER_CELL_DATA_TESTING build (device/PublicDefines.h:11): there every cell processor
// records {prevValue, matchScore, cellValue, symbol, passesThreshold} (device/HavacHls.cpp:388-399) and
// test/byCellComparator/byCellComparator.cpp:47-96 lays them next to softSsv's.  Here a second instantiation of the SAME
// kernel body (ssv_diag_kernel_traced: same tables, symbol window, skewed pairs, saturating adds, slow path) writes, after
// every single step, what each cell of a requested window went through -- decoded from the packed int16 form the kernel
// really computes in, so the trace checks the representation, not a restatement of it.  A cell that crossed the threshold
// keeps its mark until the window's hit test puts it back to score 0 (that is the design: one test per four steps), so
// the up to three cells below a crossing on the same diagonal, inside the same window of four steps, are recorded with
// `pending` set and carry no score; everything else is exact.
struct CellRecord {
    uint8_t prev;        // score of the cell above-left (the diagonal's previous cell)
    int8_t match;        // match score that was added
    uint8_t score;       // the cell's score afterwards (0 after a crossing, as device/HavacHls.cpp:384 has it)
    uint8_t hit;         // 1: the cell crossed the threshold (passesThreshold)
    uint8_t symbol;      // the sequence symbol the kernel used for the cell (0..3; 0xff under a separator)
    uint8_t pending;     // 1: an earlier cell of this diagonal crossed inside the same four-step window; prev and score carry nothing
    uint8_t zero;
    uint8_t written;     // 1 (the caller clears the buffer: a cell the kernel never visited shows 0)
};

struct SsvRare {
    uint64_t* hits;                // global queue of sort keys (hit_key)
    unsigned long long* hit_count; // records found so far (may run past capacity)
    uint64_t hit_capacity;
    int64_t col_begin;             // only hits in columns [col_begin, col_begin + col_span) are reported (the shard's own columns)
    uint64_t col_span;
    const uint32_t* abort_flag;    // optional device word: non-zero = stop
    const uint16_t* pair_mask;     // separator bitmap (boundary mode), or null
    uint32_t* tickets;             // the ticket counter of row-split launches
    uint32_t* block_flags;         // per CUT tile of this launch (see handoff slots, "items"): row blocks finished
    uint32_t* block_state;         // per cut tile: 8 x 64 words = the 2048 scores (a byte each) handed from one row block to the next
    uint32_t* fault;               // raised when a wait for a row block ran out (never expected)
    uint32_t row_bits;             // width of the row field of the sort key
    // the tiling of this launch and how it is handed out: read once per block / item (see "work distribution" below)
    int64_t first_diag; uint32_t tile_begin, ntiles;
    int64_t col_end;
    uint32_t parts_log2;           // the launch's units are dealt to 2^parts_log2 partitions: block i belongs to partition i mod 2^parts_log2
    uint32_t part_begin[9];        // partition k owns tiles [part_begin[k], part_begin[k+1]) of the launch: runs of about equal WORK (tiles at the matrix's ends are short)
    uint32_t tiles_per_item;       // >= 1; > 1 (experiments, short models): a wave walks a GROUP of that many adjacent tiles
    uint32_t single_tiles;         // ... except a partition's last single_tiles whole tiles, which are items of their own (the launch's last round)
    // ssv_resident_kernel: the launch's waves come in walk_rounds rounds of walk_slots waves (the last round: as many as it takes);
    // wave i of round r walks walk_len[r] adjacent tiles from tile walk_base[r] + i * walk_len[r] on (clipped to the launch's tiles)
    uint32_t walk_slots, walk_rounds, walk_len[kMaxWalkRounds], walk_base[kMaxWalkRounds];
    uint32_t half_tail;            // ssv_resident_kernel: 1 = the model's rows end in the first half of the last chunk, whose second half is not run
    uint32_t split_units;          // of every partition's units the LAST split_units are cut by rows (0: no tile of this launch is cut)
    uint32_t nrow_blocks;          // row blocks of a cut tile
    uint32_t ncuts, uniform_rows;  // block b < ncuts is rows [row_cut[b], row_cut[b+1]); from block ncuts on the blocks are uniform_rows rows each
    uint32_t row_cut[kMaxRowCuts + 1];
    // side buffer for what a block still has staged when it ends (see "block tails"); tails == nullptr: the atomic path
    uint64_t* tails; uint32_t* tail_counts;
    // per-cell trace (ssv_diag_kernel_traced only): one CellRecord per cell of rows [cell_row0, +cell_rows) x columns [cell_col0, +cell_cols)
    struct CellRecord* cells; int64_t cell_col0; uint32_t cell_row0, cell_rows, cell_cols;
};
typedef const __attribute__((address_space(4))) SsvRare* rare_args_t;
// (opaque: the loads through it stay where they are written instead of being hoisted to the kernel's entry.  Only valid in
// code that is INLINED into the kernel: in a function that is really called __builtin_amdgcn_kernarg_segment_ptr() is null --
// flush_full is handed the pointer, and the kernel body's item lambda is always_inline for this reason.)
__device__ __forceinline__ rare_args_t rare_args() {
    rare_args_t p = (rare_args_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

struct HitSink {
    uint64_t* stage;               // the wave's kHitStage staged records (LDS)
};

// A staged record is (row << 40 | column): two scalar instructions in the slow path; the sort key (a division by
// 12288 and three shifts) is made here, 64 records per instruction.
__device__ __forceinline__ uint64_t staged_to_key(uint64_t staged_record, uint32_t row_bits) {
    return hit_key((uint32_t)(staged_record >> 40), staged_record & ((1ull << 40) - 1ull), row_bits);
}

__device__ __forceinline__ uint32_t flush_hits(const HitSink& sink, const rare_args_t rare, uint32_t staged) {
    if (staged == 0) return 0;
    const uint32_t lane = __lane_id();
    uint64_t* const hits = rare->hits;
    const uint64_t capacity = rare->hit_capacity;
    const uint32_t row_bits = rare->row_bits;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(rare->hit_count, (unsigned long long)staged);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    for (uint32_t i = lane; i < staged; i += 64) {
        const unsigned long long idx = base + i;
        if (idx < capacity) hits[idx] = staged_to_key(sink.stage[i], row_bits);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return 0;
}

// A full stage leaves with one returning atomic (a real call: rare, and its registers stay out of the hot loop).
// (the kernarg pointer is handed in: in a function that is really called, __builtin_amdgcn_kernarg_segment_ptr() is null)
__device__ __noinline__ uint32_t flush_full(const HitSink sink, const rare_args_t rare, uint32_t staged) {
    return flush_hits(sink, rare, __builtin_amdgcn_readfirstlane(staged));
}

// The slow path's copy of one half of v_pk_add_i16 ... clamp, on the scalar unit: the cell and its match score are kept
// in the HIGH halves of 32-bit values, so that int16 saturation is int32 saturation (s_add_i32 reports the overflow).
// A crossed cell reads 0x7fffffff (no regular value does: those have a zero low half), a cell at score 0 0x80000000.
__device__ __forceinline__ int32_t scalar_sat_add(int32_t a, int32_t b) {
    const int32_t saturated = (b >> 31) ^ INT32_MAX;      // where the sum goes if it overflows: the sign of b decides
    int32_t d;
    asm("s_add_i32 %0, %1, %2\n\ts_cselect_b32 %0, %3, %0" : "=&s"(d) : "s"(a), "s"(b), "s"(saturated) : "scc");
    return d;
}

// One record of the slow path: cell (row, column), wave-uniform.  Staged as (row << 40 | column) by lane 0.
struct ShardColumns { rare_args_t rare; int64_t begin; uint64_t span; };     // read at the slow path's entry, used when a cell is reported
__device__ __forceinline__ uint32_t emit_cell(const HitSink& sink, const ShardColumns& own, uint32_t staged, uint32_t row, int64_t column) {
    // halo columns belong to the neighbouring shard: one unsigned comparison of (column - col_begin) with the span
    const rare_args_t rare = own.rare;
    if ((uint64_t)(column - own.begin) < own.span) {
        const uint64_t record = ((uint64_t)row << 40) | (uint64_t)column;
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint64_t*)sink.stage + staged * 8;
        uint64_t saved_exec;        // one lane stores: exec = lane 0 for the one instruction
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b64 %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved_exec) : "v"(addr), "v"(record) : "memory");
        staged++;
        if (staged == kHitStage) staged = flush_full(sink, rare, staged);
    }
    return staged;
}

// both steps' match words of one register: one ds_read_b64, table of step pair P, entry `code_addr`
template <int P>
__device__ __forceinline__ u32x2 match_words(uint32_t code_addr, int extra = 0) {
    return *(lds_words_t)(uintptr_t)(code_addr + P * kPairStride + extra);
}
// Resident tables (ssv_resident_body: every chunk of a short model has its own table in LDS, kTableBytes apart, built once per
// workgroup).  A window entry carries the base of the table of the chunk it was EXPANDED in; in the chunk after that -- where it
// has slid into the window's lower half, entries 0..15 -- it is used with the next table, one kTableBytes further: a constant
// that goes into the read's offset field, since an entry's index is static in the unrolled chunk.
template <bool Resident>
__device__ __forceinline__ constexpr int slid_offset(int entry_index) { return Resident && entry_index < 16 ? kTableBytes : 0; }

// A wave-uniform 32-bit value the optimiser cannot see through, kept in an SGPR.  Two uses: (a) 64-bit comparisons of
// uniform values are done on the VALU (there is no s_cmp_lt_i64) unless they are taken apart into 32-bit halves that the
// optimiser cannot put together again; (b) a uniform flag that lives across the chunk loop is otherwise kept as a lane
// mask and converted back and forth through a VGPR (v_cndmask + v_cmp per use).
__device__ __forceinline__ uint32_t opaque_uniform(uint32_t x) {
    x = __builtin_amdgcn_readfirstlane(x);      // a copy where the value is in an SGPR already
    asm("" : "+s"(x));
    return x;
}
// The same, evaluated afresh at every use: a uniform flag tested in several basic blocks is otherwise turned into ONE lane mask
// whose copies and inversions go through a VGPR (v_cndmask + v_cmp per block); tested afresh it is s_cmp + s_cbranch_scc.
__device__ __forceinline__ uint32_t fresh_uniform(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
// v clamped to [0, 4096], on the scalar unit
__device__ __forceinline__ int32_t clamp_to_4096(int64_t v) {
    const int32_t hi = (int32_t)opaque_uniform((uint32_t)((uint64_t)v >> 32));
    const uint32_t lo = opaque_uniform((uint32_t)v);
    return hi < 0 ? 0 : ((hi > 0 || lo > 4096u) ? 4096 : (int32_t)lo);
}

// The lane number, computed where it is needed (three instructions) and opaque to the optimiser, which would otherwise
// compute it once and keep it -- and what is derived from it -- in VGPRs through the whole kernel.
__device__ __forceinline__ uint32_t fresh_lane() {
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}
// What a lane needs once per chunk for the table build and the loads of the next chunk lives in LDS, as four arrays of 64
// words read with ds_read_addtid_b32 (address = M0 + offset + 4 * lane: no address register either):
// {selector of its second match word, LDS offset of its four table entries, byte offset of its step pair in a chunk's rows,
// 8 * lane = byte offset of its 32 symbols in the wave's 512 packed bytes}.
struct LaneWords { uint32_t sel_second, entries, row_offset, lane8; };
__device__ __forceinline__ LaneWords read_lane_words(uint32_t lds_address) {
    LaneWords w;
    // (s_nop: an SALU write of M0 needs one wait state before an add-TID LDS instruction reads it; nothing inserts it inside an asm)
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "ds_read_addtid_b32 %0 offset:0\n\tds_read_addtid_b32 %1 offset:256\n\t"
                 "ds_read_addtid_b32 %2 offset:512\n\tds_read_addtid_b32 %3 offset:768\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=v"(w.sel_second), "=v"(w.entries), "=v"(w.row_offset), "=v"(w.lane8) : "s"(lds_address) : "m0", "memory");
    return w;
}

// ---- the symbol window, expanded as the windows reach it --------------------------------------------------------
// A chunk of 32 steps looks at 64 symbols per lane = 32 symbol pairs; C[k] is the LDS address of the table entry of
// pair k (code * 8 | table base).  Pairs 0..15 are the upper half of the previous chunk's window; pairs 16..31 come from
// the 8 bytes loaded for this chunk and are expanded entry by entry: window Q (steps 4Q..4Q+3) uses entries 2Q..2Q+16
// only, so 17 of the 32 addresses (plus the four prepared words) are alive at a time -- the registers that let both
// score sets stay alive at six waves per SIMD.
constexpr uint32_t kFlagSpecial = 2u;
constexpr int kFlagSafeShift = 8;       // bits 8 .. 15 of the flag word: window Q of the chunk may look for hits at its end only (ssv_prepare_model)
struct LazySymbols {
    uint32_t even[2], odd[2];   // per packed word (pairs 16..23, 24..31): code*8 of its even / odd pairs, one per byte
    uint32_t separators;        // bit K: pair 16+K is a separator pair (boundary mode); 0 otherwise
    uint32_t table_base;
    uint32_t special;           // wave-uniform flag word (a 32-bit SGPR, not a lane mask).  kFlagSpecial: some position of the wave lies outside [0, N), or a
                                // separator is present; bits kFlagSafeShift + Q (set by the chunk loop for the chunk it enters): window Q may test for hits at
                                // its end only.  ONE word for both: every window entry's asm takes it as an "s" operand, and with two such words alive through
                                // the windows hipcc's allocator once moved one into a VGPR ("illegal VGPR to SGPR copy")
    int32_t valid_lo, valid_hi; // wave-uniform: positions relative to the wave's first that lie inside [0, N): [valid_lo, valid_hi)
};

__device__ __forceinline__ void prepare_symbols(LazySymbols& z, uint32_t lo, uint32_t hi) {
    // pair n of a word = its bits [4n+3:4n]; code*8 = bits [6:3] of a byte: even pairs (w << 3), odd pairs (w >> 1)
    z.even[0] = (lo << 3) & 0x78787878u; z.odd[0] = (lo >> 1) & 0x78787878u;
    z.even[1] = (hi << 3) & 0x78787878u; z.odd[1] = (hi >> 1) & 0x78787878u;
}

template <int BYTE>
__device__ __forceinline__ uint32_t or_byte(uint32_t base, uint32_t word) {   // base | ((word >> 8*BYTE) & 0xff), one SDWA instruction
    uint32_t out;
    if constexpr (BYTE == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(out) : "s"(base), "v"(word));
    if constexpr (BYTE == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(out) : "s"(base), "v"(word));
    if constexpr (BYTE == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(out) : "s"(base), "v"(word));
    if constexpr (BYTE == 3) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(out) : "s"(base), "v"(word));
    return out;
}

template <int K>
__device__ __forceinline__ void expand_entry(uint32_t (&C)[32], const LazySymbols& z) {
    constexpr int n = K & 7;
    uint32_t entry = or_byte<n / 2>(z.table_base, (n & 1) ? z.odd[K / 8] : z.even[K / 8]);
    if (fresh_uniform(z.special) & kFlagSpecial) {
        // positions outside [0, N) and separator pairs use the 17th entry of the tables
        const uint32_t lane32 = fresh_lane() * 32u;           // computed here, in the rare chunks that need it (nothing that
                                                                // depends on the lane number is kept in a VGPR through the windows)
        const int32_t q = (int32_t)lane32 + 2 * K;              // the pair's positions q and q+1 are in or out together
        if (q < z.valid_lo || q >= z.valid_hi || ((z.separators >> K) & 1u)) entry = z.table_base + kOutsideCode;
    }
    C[16 + K] = entry;
}

// what window Q needs beyond what window Q-1 had: entries 2Q+15 and 2Q+16 (window 0: entry 16)
template <int Q>
__device__ __forceinline__ void expand_for_window(uint32_t (&C)[32], const LazySymbols& z) {
    if constexpr (Q == 0) {
        expand_entry<0>(C, z);
    } else {
        expand_entry<2 * Q - 1>(C, z);
        expand_entry<2 * Q>(C, z);
    }
}

template <int... K>
__device__ __forceinline__ void expand_all(uint32_t (&C)[32], const LazySymbols& z, std::integer_sequence<int, K...>) {
    (expand_entry<K>(C, z), ...);
}

// One step of one register, traced: `before` and `after` are the register around the add of match word `mword`, `entry` the
// table entry the word came from (table_base | code * 8, or the outside entry).  At step t the register's low cell is
// (row t, column c) and its high cell (row t - 1, column c), c = wave_diag0 + 32 lane + 2 reg + t.
// A REAL call (noinline): the traced body then stays about as big as the production one and is inlined the same way; the
// window comes as arguments because a called function cannot read the kernarg segment (see rare_args).
struct TraceWindow { CellRecord* cells; int64_t col0; uint32_t row0, nrows, ncols; };
__device__ __noinline__ void trace_cells_of_step(TraceWindow win, uint32_t second_of_pair, uint32_t reg, uint32_t t, int64_t wave_diag0,
                                                 uint32_t before, uint32_t mword, uint32_t after, uint32_t code) {
    const int64_t column = wave_diag0 + (int64_t)(32u * __lane_id() + 2u * reg) + (int64_t)t;
    const uint64_t dc = (uint64_t)(column - win.col0);
    const uint8_t symbol = code > 15u ? 0xffu : (uint8_t)(second_of_pair ? code >> 2 : code & 3u);   // code: the symbol pair (a | b << 2); 16: outside / separator
    for (uint32_t half = 0; half < 2; half++) {
        const uint32_t dr = t - half - win.row0;                      // the high cell runs one row behind
        if (dr < win.nrows && dc < (uint64_t)win.ncols) {
            const uint32_t b = (before >> (16 * half)) & 0xffffu, a = (after >> (16 * half)) & 0xffffu;
            CellRecord rec;
            rec.pending = (b & 0xffu) != 0;                           // still marked from an earlier step of this window
            rec.hit = !rec.pending && (a & 0xffu) != 0;
            rec.prev = rec.pending ? 0 : (uint8_t)((b ^ 0x8000u) >> 8);
            rec.score = (rec.pending || rec.hit) ? 0 : (uint8_t)((a ^ 0x8000u) >> 8);
            rec.match = (int8_t)(uint8_t)(mword >> (8 + 16 * half));
            rec.symbol = symbol;
            rec.zero = 0;
            rec.written = 1;
            win.cells[(size_t)dr * win.ncols + (size_t)dc] = rec;
        }
    }
}
__device__ __forceinline__ TraceWindow trace_window() {          // inlined into the kernel: reads the kernarg segment
    const rare_args_t rare = rare_args();
    return TraceWindow{rare->cells, rare->cell_col0, rare->cell_row0, rare->cell_rows, rare->cell_cols};
}
template <bool SecondOfPair>
__device__ __forceinline__ void trace_step(const TraceWindow& win, uint32_t reg, uint32_t t, int64_t wave_diag0, uint32_t before,
                                           uint32_t mword, uint32_t after, uint32_t entry, uint32_t table_base) {
    trace_cells_of_step(win, SecondOfPair ? 1u : 0u, reg, t, wave_diag0, before, mword, after, (entry - table_base) >> 3);
}

// ---- the slow path ------------------------------------------------------------------------------------------------
// What the slow path of a window needs of register I of lane l: the scores the window started from and the LDS
// addresses of the two table entries its four match words came from.  Selected by a chain of scalar compares on the
// register number, so that the replay exists once per window and not once per register.  Written as asm: left as
// plain C++, hipcc folds the chain into ONE dynamically indexed access, for which the 16 score registers (and the 32
// window registers) have to live in consecutive VGPRs -- it then spills them as whole tuples inside the hot loop.
template <int Q, int I>
__device__ __forceinline__ void window_inputs(uint32_t l, const uint32_t (&cur)[kRegs], const uint32_t (&nxt)[kRegs],
                                              const uint32_t (&C)[32], uint32_t& s, uint32_t& now, uint32_t& entry0,
                                              uint32_t& entry1) {
    asm volatile("v_readlane_b32 %0, %4, %8\n\tv_readlane_b32 %1, %5, %8\n\tv_readlane_b32 %2, %6, %8\n\tv_readlane_b32 %3, %7, %8"
                 : "=&s"(s), "=&s"(now), "=&s"(entry0), "=&s"(entry1)
                 : "v"(cur[I]), "v"(nxt[I]), "v"(C[2 * Q + I]), "v"(C[2 * Q + 1 + I]), "s"(l));
}

// Lane l of score register r (0..15, wave-uniform) := s.  ONE asm statement that names all 16 registers and branches
// inside (a binary tree on r: four compares): written as sixteen conditional C++ statements, the merge of the sixteen paths makes hipcc treat the register
// set as one 512-bit value that it copies and spills as a whole in the hot loop (1.2 KB of scratch).
__device__ __forceinline__ void write_score_lane(uint32_t (&n)[kRegs], uint32_t s, uint32_t l, uint32_t r) {
    static_assert(kRegs == 16, "the asm below names sixteen registers");
    asm volatile("s_mov_b32 m0, %[l]\n\t"
                 "s_cmp_ge_u32 %[r], 8\n\ts_cbranch_scc1 .Lhavac_w8_16_%=\n\t"
                 "s_cmp_ge_u32 %[r], 4\n\ts_cbranch_scc1 .Lhavac_w4_8_%=\n\t"
                 "s_cmp_ge_u32 %[r], 2\n\ts_cbranch_scc1 .Lhavac_w2_4_%=\n\t"
                 "s_cmp_ge_u32 %[r], 1\n\ts_cbranch_scc1 .Lhavac_w1_2_%=\n\t"
                 "v_writelane_b32 %0, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w1_2_%=:\n\t"
                 "v_writelane_b32 %1, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w2_4_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 3\n\ts_cbranch_scc1 .Lhavac_w3_4_%=\n\t"
                 "v_writelane_b32 %2, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w3_4_%=:\n\t"
                 "v_writelane_b32 %3, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w4_8_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 6\n\ts_cbranch_scc1 .Lhavac_w6_8_%=\n\t"
                 "s_cmp_ge_u32 %[r], 5\n\ts_cbranch_scc1 .Lhavac_w5_6_%=\n\t"
                 "v_writelane_b32 %4, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w5_6_%=:\n\t"
                 "v_writelane_b32 %5, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w6_8_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 7\n\ts_cbranch_scc1 .Lhavac_w7_8_%=\n\t"
                 "v_writelane_b32 %6, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w7_8_%=:\n\t"
                 "v_writelane_b32 %7, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w8_16_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 12\n\ts_cbranch_scc1 .Lhavac_w12_16_%=\n\t"
                 "s_cmp_ge_u32 %[r], 10\n\ts_cbranch_scc1 .Lhavac_w10_12_%=\n\t"
                 "s_cmp_ge_u32 %[r], 9\n\ts_cbranch_scc1 .Lhavac_w9_10_%=\n\t"
                 "v_writelane_b32 %8, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w9_10_%=:\n\t"
                 "v_writelane_b32 %9, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w10_12_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 11\n\ts_cbranch_scc1 .Lhavac_w11_12_%=\n\t"
                 "v_writelane_b32 %10, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w11_12_%=:\n\t"
                 "v_writelane_b32 %11, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w12_16_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 14\n\ts_cbranch_scc1 .Lhavac_w14_16_%=\n\t"
                 "s_cmp_ge_u32 %[r], 13\n\ts_cbranch_scc1 .Lhavac_w13_14_%=\n\t"
                 "v_writelane_b32 %12, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w13_14_%=:\n\t"
                 "v_writelane_b32 %13, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w14_16_%=:\n\t"
                 "s_cmp_ge_u32 %[r], 15\n\ts_cbranch_scc1 .Lhavac_w15_16_%=\n\t"
                 "v_writelane_b32 %14, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_w15_16_%=:\n\t"
                 "v_writelane_b32 %15, %[s], m0\n\ts_branch .Lhavac_wend_%=\n"
                 ".Lhavac_wend_%=:"
                 : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]),
                   "+v"(n[8]), "+v"(n[9]), "+v"(n[10]), "+v"(n[11]), "+v"(n[12]), "+v"(n[13]), "+v"(n[14]), "+v"(n[15])
                 : [s] "s"(s), [l] "s"(l), [r] "s"(r)
                 : "m0", "scc");
}

// ONE marked register of lane l among registers [Lo, Hi) (there is one): halve the range by OR-ing the lower half's
// registers (two to four vector instructions on static registers) and looking at lane l of the result; at the leaf, the
// values the replay needs of that register, read with static register names.  Four levels, ~25 instructions, where
// reading the lane of all 16 registers took 64.
template <int Q, int Lo, int Hi>
__device__ __forceinline__ void find_marked(uint32_t l, const uint32_t (&cur)[kRegs], const uint32_t (&nxt)[kRegs],
                                            const uint32_t (&C)[32], uint32_t& r, uint32_t& s, uint32_t& now,
                                            uint32_t& entry0, uint32_t& entry1) {
    if constexpr (Hi - Lo == 1) {
        r = Lo;
        window_inputs<Q, Lo>(l, cur, nxt, C, s, now, entry0, entry1);
    } else {
        constexpr int Mid = (Lo + Hi) / 2;
        uint32_t lower = nxt[Lo];
#pragma unroll
        for (int i = Lo + 1; i < Mid; i++) lower |= nxt[i];
        if (__builtin_amdgcn_readlane(lower, l) & kCrossedBits) find_marked<Q, Lo, Mid>(l, cur, nxt, C, r, s, now, entry0, entry1);
        else find_marked<Q, Mid, Hi>(l, cur, nxt, C, r, s, now, entry0, entry1);
    }
}

#ifdef HAVAC_SLOW_CLOCKS
// experiments only (tools/slow_path_clocks.py): shader-clock cycles (s_memtime) the waves spent inside the slow path, and its entries;
// summed per workgroup in LDS (one atomic pair per entry on one global word would be the slowest thing in the kernel), per launch here
__device__ unsigned long long g_slow_clocks[4];      // [2], [3]: s_memtime and s_memrealtime (100 MHz) over the workgroups' lives, summed: the shader clock the launch really saw
__shared__ unsigned long long s_slow_clocks[4];      // [2], [3] here: the two clocks at the workgroup's start
#endif
// `marked` (per lane) = OR of the score registers after NSTEPS (2 or 4) steps of window Q.  For every lane and register
// that shows a mark: take the register's NSTEPS steps again on the scalar unit from `cur` (the scores the window started
// from), report the crossings of steps >= report_from (an unsafe window has reported those of its first two steps at
// its middle already), restart crossed cells from score 0 on the way, and write the exact result over `nxt`.
// At step t the low cell of register r of lane l is (row t, column c) and the high cell (row t - 1, column c),
// c = wave_diag0 + 32 l + 2 r + t.
template <bool Resident, int Q, int NSTEPS, int... I>
__device__ __forceinline__ void window_slow(const uint32_t (&cur)[kRegs], uint32_t (&nxt)[kRegs], const uint32_t (&C)[32],
                                            unsigned long long lanes, int report_from, const HitSink& sink, uint32_t& staged,
                                            uint32_t step0, int64_t wave_diag0, std::integer_sequence<int, I...>) {
#if defined(HAVAC_SLOW_CLOCKS) && HAVAC_SLOW_CLOCKS > 1      // (-DHAVAC_SLOW_CLOCKS=1: only the launch's shader clock, the kernel runs at its usual speed)
    const uint64_t slow_t0 = __builtin_readcyclecounter();
#endif
    // the shard's columns, from the kernarg segment: the loads are issued here and are back long before a cell is reported
    ShardColumns own;
    own.rare = rare_args();
    own.begin = own.rare->col_begin;
    own.span = own.rare->col_span;
    do {                                  // `lanes`: the lanes that show a mark (the hit test's own comparison, handed in)
        do {
            const uint32_t l = (uint32_t)__builtin_ctzll(lanes);
            lanes &= lanes - 1;
            // one marked register of this lane (a second one, rare, is found by the look after the lanes)
            uint32_t r = 0, s = 0, now = 0, entry0 = 0, entry1 = 0;      // register; scores at the window's start and now; table entries
            find_marked<Q, 0, kRegs>(l, cur, nxt, C, r, s, now, entry0, entry1);
            // the match words again, from the lane's table entries (uniform addresses: every lane reads the same)
            uint32_t w[kWindowSteps] = {0, 0, 0, 0};
            if constexpr (Resident) {      // (the register's number is only known here: the slid entries' constant, see slid_offset)
                entry0 += 2 * Q + r < 16 ? (uint32_t)kTableBytes : 0u;
                entry1 += 2 * Q + 1 + r < 16 ? (uint32_t)kTableBytes : 0u;
            }
            const u32x2 wa = match_words<2 * Q>(entry0);
            w[0] = (uint32_t)__builtin_amdgcn_readfirstlane(wa.x);
            w[1] = (uint32_t)__builtin_amdgcn_readfirstlane(wa.y);
            if constexpr (NSTEPS > 2) {
                const u32x2 wb = match_words<2 * Q + 1>(entry1);
                w[2] = (uint32_t)__builtin_amdgcn_readfirstlane(wb.x);
                w[3] = (uint32_t)__builtin_amdgcn_readfirstlane(wb.y);
            }
            const uint32_t t0 = step0 + kWindowSteps * Q;
            const int64_t column0 = wave_diag0 + (int64_t)(32 * l + 2 * r) + (int64_t)t0;
            // Only a cell that shows a mark crossed in these steps (a mark outlives them: that is what `safe` and the
            // middle test are for); the other cell of the register is exact as it stands.
            int32_t low = (int32_t)(s << 16), high = (int32_t)(s & 0xffff0000u);     // the register's two cells
            // The replay only NOTES which steps crossed (bit k: the low cell at step k; bit 4 + k: the high cell) and has no
            // branch in it; the records are made afterwards, in one place (eight inlined copies of the reporting code, with
            // everything the optimiser hoisted out of them, cost the slow path 16 SGPRs and their spills).
            uint32_t crossed = 0;
            if (now & 1u) {
#pragma unroll
                for (int k = 0; k < NSTEPS; k++) {
                    low = scalar_sat_add(low, (int32_t)(w[k] << 16));
                    const bool hit = low == INT32_MAX;            // crossed at (row t0 + k, column column0 + k)
                    crossed |= hit ? (1u << k) : 0u;
                    low = hit ? INT32_MIN : low;
                }
                now = (now & 0xffff0000u) | ((uint32_t)low >> 16);
            }
            if (now & 0x10000u) {
#pragma unroll
                for (int k = 0; k < NSTEPS; k++) {
                    high = scalar_sat_add(high, (int32_t)(w[k] & 0xffff0000u));
                    const bool hit = high == INT32_MAX;           // the high cell: one row behind, the same column
                    crossed |= hit ? (16u << k) : 0u;
                    high = hit ? INT32_MIN : high;
                }
                now = (now & 0xffffu) | ((uint32_t)high & 0xffff0000u);
            }
            crossed &= ((0xfu << report_from) & 0xfu) * 0x11u;      // steps before report_from were reported at the window's middle
            while (crossed) {
                const uint32_t bit = (uint32_t)__builtin_ctz(crossed);
                crossed &= crossed - 1;
                const uint32_t k = bit & 3u, behind = bit >> 2;
                staged = emit_cell(sink, own, staged, t0 + k - behind, column0 + k);
            }
            s = now;
            write_score_lane(nxt, s, l, r);
        } while (lanes);
        uint32_t marked = 0;
#pragma unroll
        for (int i = 0; i < kRegs; i++) marked |= nxt[i];
        lanes = __ballot((marked & kCrossedBits) != 0);
    } while (lanes);
#if defined(HAVAC_SLOW_CLOCKS) && HAVAC_SLOW_CLOCKS > 1      // (-DHAVAC_SLOW_CLOCKS=1: only the launch's shader clock, the kernel runs at its usual speed)
    if (fresh_lane() == 0) { atomicAdd(&s_slow_clocks[0], (unsigned long long)(__builtin_readcyclecounter() - slow_t0)); atomicAdd(&s_slow_clocks[1], 1ull); }
#endif
}

// Steps 4Q .. 4Q+3 of the chunk.  `cur` is left untouched (the first add is not in place) and holds the scores the window
// started from; `nxt` receives the scores after the four steps; the next window swaps the two sets.  One hit test at
// the end where the chunk flags (ssv_prepare_model) say that is exact (`safe`, wave-uniform), else one more in the middle.
template <bool Trace, bool Resident, int Q, int... I>
__device__ __forceinline__ void step_window(const uint32_t (&cur)[kRegs], uint32_t (&nxt)[kRegs], const uint32_t (&C)[32],
                                            uint32_t safe, const HitSink& sink, uint32_t& staged, uint32_t step0,
                                            int64_t wave_diag0, uint32_t table_base, std::integer_sequence<int, I...> regs) {
    static_assert(sizeof...(I) == kRegs, "one index per score register");
    constexpr int H = HAVAC_BATCH;          // match words are read for eight registers at a time: 16 VGPRs in flight, not 32
    constexpr int NB = kRegs / H;
    TraceWindow win{};
    if constexpr (Trace) win = trace_window();
    // steps 4Q, 4Q+1
#pragma unroll
    for (int h = 0; h < NB; h++) {
        u32x2 m[H];
        __builtin_amdgcn_sched_barrier(0);      // reads stay behind the adds before them: hoisted, they cost registers the kernel does not have
#pragma unroll
        for (int i = 0; i < H; i++) m[i] = match_words<2 * Q>(C[2 * Q + h * H + i], slid_offset<Resident>(2 * Q + h * H + i));
        __builtin_amdgcn_sched_barrier(0);      // all eight reads in flight before the first add waits (hipcc otherwise staggers them)
#pragma unroll
        for (int i = 0; i < H; i++) asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(nxt[h * H + i]) : "v"(cur[h * H + i]), "v"(m[i].x));
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++)
                trace_step<false>(win, h * H + i, step0 + 4 * Q, wave_diag0, cur[h * H + i], m[i].x, nxt[h * H + i], C[2 * Q + h * H + i], table_base);
        }
        uint32_t mid[H];
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++) mid[i] = nxt[h * H + i];
        }
#pragma unroll
        for (int i = 0; i < H; i++) asm("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[h * H + i]) : "v"(m[i].y));
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++)
                trace_step<true>(win, h * H + i, step0 + 4 * Q + 1, wave_diag0, mid[i], m[i].y, nxt[h * H + i], C[2 * Q + h * H + i], table_base);
        }
    }
    if (!(fresh_uniform(safe) & (1u << (kFlagSafeShift + Q)))) {
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < kRegs; i++) any |= nxt[i];
        if (const unsigned long long lanes = __ballot((any & kCrossedBits) != 0); __builtin_expect(lanes != 0, 0))
            window_slow<Resident, Q, 2>(cur, nxt, C, lanes, 0, sink, staged, step0, wave_diag0, regs);
    }
    // steps 4Q+2, 4Q+3
#pragma unroll
    for (int h = 0; h < NB; h++) {
        u32x2 m[H];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < H; i++) m[i] = match_words<2 * Q + 1>(C[2 * Q + 1 + h * H + i], slid_offset<Resident>(2 * Q + 1 + h * H + i));
        __builtin_amdgcn_sched_barrier(0);
        uint32_t mid[H];
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++) mid[i] = nxt[h * H + i];
        }
#pragma unroll
        for (int i = 0; i < H; i++) asm("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[h * H + i]) : "v"(m[i].x));
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++) {
                trace_step<false>(win, h * H + i, step0 + 4 * Q + 2, wave_diag0, mid[i], m[i].x, nxt[h * H + i], C[2 * Q + 1 + h * H + i], table_base);
                mid[i] = nxt[h * H + i];
            }
        }
#pragma unroll
        for (int i = 0; i < H; i++) asm("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[h * H + i]) : "v"(m[i].y));
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++)
                trace_step<true>(win, h * H + i, step0 + 4 * Q + 3, wave_diag0, mid[i], m[i].y, nxt[h * H + i], C[2 * Q + 1 + h * H + i], table_base);
        }
    }
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < kRegs; i++) any |= nxt[i];
    if (const unsigned long long lanes = __ballot((any & kCrossedBits) != 0); __builtin_expect(lanes != 0, 0))
        window_slow<Resident, Q, kWindowSteps>(cur, nxt, C, lanes, (int)(2u - 2u * ((safe >> (kFlagSafeShift + Q)) & 1u)), sink, staged, step0, wave_diag0, regs);
}

// ONE step with the tables of step pair 2 * PQ (PQ = 0: the step behind the model's last chunk, after the window has slid: the high
// cells run one row behind and still owe the last row; the low cells add a padding row, which scores 0.  PQ = 4, round 5: the same
// step in the MIDDLE of the last chunk, where a short model's rows end in the chunk's first half -- `step0` is then the chunk's
// first step, as for a window).
template <bool Trace, bool Resident, int PQ, int... I>
__device__ __forceinline__ void step_last(const uint32_t (&cur)[kRegs], uint32_t (&nxt)[kRegs], const uint32_t (&C)[32],
                                          const HitSink& sink, uint32_t& staged, uint32_t step0, int64_t wave_diag0,
                                          uint32_t table_base, std::integer_sequence<int, I...> regs) {
    constexpr int H = kRegs / 2;
    TraceWindow win{};
    if constexpr (Trace) win = trace_window();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u32x2 m[H];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < H; i++) m[i] = match_words<2 * PQ>(C[2 * PQ + h * H + i], slid_offset<Resident>(2 * PQ + h * H + i));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < H; i++) asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(nxt[h * H + i]) : "v"(cur[h * H + i]), "v"(m[i].x));
        if constexpr (Trace) {
#pragma unroll
            for (int i = 0; i < H; i++)
                trace_step<false>(win, h * H + i, step0 + kWindowSteps * PQ, wave_diag0, cur[h * H + i], m[i].x, nxt[h * H + i], C[2 * PQ + h * H + i], table_base);
        }
    }
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < kRegs; i++) any |= nxt[i];
    if (const unsigned long long lanes = __ballot((any & kCrossedBits) != 0); __builtin_expect(lanes != 0, 0))
        window_slow<Resident, PQ, 1>(cur, nxt, C, lanes, 0, sink, staged, step0, wave_diag0, regs);
}

// windows Q... of a chunk, in order (all eight, or -- round 5, the resident-table kernel -- the first four and the last four as two
// calls, the second of which is not made where a short model ends in the chunk's first half); an even number of them starting at an
// even one: the scores are in `a` on entry and in `a` again on exit
template <bool Trace, bool Resident, int... Q>
__device__ __forceinline__ void step_windows(uint32_t (&a)[kRegs], uint32_t (&b)[kRegs], uint32_t (&C)[32],
                                             const LazySymbols& z, uint32_t safe, const HitSink& sink, uint32_t& staged,
                                             uint32_t step0, int64_t wave_diag0, std::integer_sequence<int, Q...>) {
    static_assert(sizeof...(Q) % 2 == 0 && sizeof...(Q) * kWindowSteps <= kChunkRows, "an even number of windows");
    constexpr int last = std::max({Q...});
    ((expand_for_window<Q>(C, z),
      (Q % 2 == 0 ? step_window<Trace, Resident, Q>(a, b, C, safe, sink, staged, step0, wave_diag0, z.table_base, std::make_integer_sequence<int, kRegs>{})
                  : step_window<Trace, Resident, Q>(b, a, C, safe, sink, staged, step0, wave_diag0, z.table_base, std::make_integer_sequence<int, kRegs>{}))), ...);
    if constexpr (last == kChunkRows / kWindowSteps - 1) expand_entry<15>(C, z);      // entry 31: entry 15 of the next chunk
}

// ---- block tails ----------------------------------------------------------------------------------------------------
// (the end of every SSV kernel body: `stage` = the wave's staged records, `staged` of them, wave-uniform)
__device__ __forceinline__ void leave_block(const uint64_t* const stage, const uint32_t staged, const uint32_t wave) {
#ifdef HAVAC_SLOW_CLOCKS
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_slow_clocks[1]) { atomicAdd(&g_slow_clocks[0], s_slow_clocks[0]); atomicAdd(&g_slow_clocks[1], s_slow_clocks[1]); }
        atomicAdd(&g_slow_clocks[2], (unsigned long long)__builtin_readcyclecounter() - s_slow_clocks[2]);
        atomicAdd(&g_slow_clocks[3], (unsigned long long)wall_clock64() - s_slow_clocks[3]);
    }
#endif
    // What is still staged when the block ends.  A returning atomic on the one counter word is the obvious way out, and
    // for short models the wrong one: the word sustains ~90 returning atomics per microsecond chip-wide, a launch of
    // one-chunk tiles ends 100+ blocks per microsecond, and every one of them waits ~2 us for its answer.  So the block
    // writes its tail -- up to kTailSlots records -- with plain stores into ITS OWN slots of a side buffer, notes the count
    // and is gone; ssv_gather_tails, a small kernel behind this one, moves the tails into the queue with one atomic per
    // 256 blocks.  A tail that does not fit (or a launch too big for a side buffer) takes the atomic, once per block.
    __shared__ uint32_t block_staged[kWavesPerBlock];
    __shared__ unsigned long long block_base;
    const uint32_t lane_again = __lane_id();     // not kept through the items: two instructions here
    if (lane_again == 0) block_staged[wave] = staged;
    __syncthreads();
    uint32_t total = 0, before_me = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; w++) { total += block_staged[w]; before_me += (uint32_t)w < wave ? block_staged[w] : 0u; }
    const rare_args_t rare = rare_args();
    uint64_t* const tails = rare->tails;
    if (tails && total <= (uint32_t)kTailSlots) {
        if (wave == 0 && lane_again == 0) rare->tail_counts[blockIdx.x] = total;
        if (staged) {
            const uint32_t row_bits = rare->row_bits;
            uint64_t* const mine = tails + (size_t)blockIdx.x * kTailSlots + before_me;
            for (uint32_t i = lane_again; i < staged; i += 64) mine[i] = staged_to_key(stage[i], row_bits);
        }
        return;
    }
    if (wave == 0 && lane_again == 0) {      // (not threadIdx.x: it would sit in a VGPR -- in scratch -- from the kernel's first instruction on)
        if (tails) rare->tail_counts[blockIdx.x] = 0;
        block_base = total ? atomicAdd(rare->hit_count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (staged) {
        uint64_t* const hits = rare->hits;
        const uint64_t capacity = rare->hit_capacity;
        const uint32_t row_bits = rare->row_bits;
        const unsigned long long base = block_base + before_me;
        for (uint32_t i = lane_again; i < staged; i += 64)
            if (base + i < capacity) hits[base + i] = staged_to_key(stage[i], row_bits);
    }
}

// selector of the match word of symbol a: bytes [0x0c, a, 0x0c, 4 + a]; with v_perm(S0 = row t-1, S1 = row t) it
// yields (row_t[a] << 8) | (row_tm1[a] << 24)
__device__ __forceinline__ uint32_t word_selector(uint32_t a) { return 0x040c000cu + a * 0x01000100u; }

// ---- work distribution ------------------------------------------------------------------------------------------------
// An item is a tile (2048 diagonals) or ONE ROW BLOCK of a tile: one item per wave, four per block, the blocks handed
// out by the hardware's block scheduler.  Whole tiles balance badly at the end of a launch when they are tall (C3 as
// stated: 5,130 tiles of 503,329 rows -- not even one round of wave slots; C5: eight rounds of 4 ms tiles), so the tiles
// that run LAST are cut by rows (the host decides which and where: havac_dev.hip, plan_row_cuts).  The wave that finishes
// rows [b0, b1) of a cut tile leaves its 16 score registers in global memory and raises the tile's block count; the
// wave with the next row block of that tile picks them up.  See "items" in ssv_diag_body for the order of the items and
// why a wait for a row block always ends.
constexpr uint32_t kWholeTile = 0xffffffffu;  // run_item: not a row block, the whole tile
constexpr int kBlocksPerCu = 4 * HAVAC_WAVES_PER_SIMD / kWavesPerBlock;   // 24 waves per CU = 6 waves per SIMD
constexpr uint32_t kHandoffSpins = 1u << 26;   // x ~1 us: a minute, far beyond any row block (dense-hit models take ~1 s each)

// 6 waves per SIMD: 80 VGPRs = cur 16 + nxt 16 + 16 match words in flight + ~19 window addresses + 4 prepared symbol words + a
// few temporaries.  Nothing else is kept in a VGPR through the windows: no lane number (LaneWords, fresh_lane), no
// SGPR spills (SsvRare).  The sixth wave is worth 3 % (C2 kernel 1.93 -> 1.87 ms); a seventh (72 VGPRs) needs batches of
// four match-word pairs and spills in the chunk epilogue.
template <bool Trace>
__device__ __forceinline__ void ssv_diag_body(const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                                              const uint32_t* __restrict__ safe_chunks, const int64_t nsymbols,
                                              const uint32_t nrows_padded) {
    __shared__ WaveLds wave_lds[kWavesPerBlock];
#ifdef HAVAC_SLOW_CLOCKS
    if (threadIdx.x == 0) { s_slow_clocks[0] = s_slow_clocks[1] = 0; s_slow_clocks[2] = __builtin_readcyclecounter(); s_slow_clocks[3] = wall_clock64(); }
    __syncthreads();
#endif

    const uint32_t lane = threadIdx.x & 63;
    // is a separator mask in use?  As a 32-bit SGPR flag tested afresh (fresh_uniform), not a lane mask that is copied through -- and
    // once spilled from -- a VGPR
    const uint32_t has_mask = opaque_uniform(rare_args()->pair_mask != nullptr ? 1u : 0u);
    // readfirstlane: everything derived from the tile index is wave-uniform (SALU branches, scalar address math)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds* const lds = &wave_lds[wave];
    const HitSink sink{lds->stage};
    uint32_t staged = 0;          // wave-uniform

    // LDS byte address of this wave's tables (a multiple of 128: code*8 is OR-ed into it)
    const uint32_t table_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds->table);
    // this lane's share of the table build: the four entries (a, b) with b = lane & 3 of step pair lane >> 2
    const uint32_t my_pair = lane >> 2, my_b = lane & 3;
    // (see LaneWords; every wave writes the same values and reads back its own)
    __shared__ uint32_t lane_words[4][64];
    lane_words[0][lane] = word_selector(my_b);
    lane_words[1][lane] = my_pair * kPairStride + my_b * 32;
    lane_words[2][lane] = my_pair * 8;
    lane_words[3][lane] = lane * 8;
    const uint32_t lane_words_address = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lane_words[0][0]);
    if (my_b == 0) {      // the "outside the matrix" entry of each table never changes
        // Without separators it scores 0: a cell outside the matrix keeps what it has (0 before a diagonal enters at
        // column 0; behind the last column nothing is reported any more) and a mark is never lost there.  With a
        // separator mask the same entry scores -128: two of them reset every diagonal through a separator pair.
        const uint32_t outside = has_mask ? kOutsideReset : kOutsideNeutral;
        *(lds_words_out_t)(uintptr_t)(table_base + my_pair * kPairStride + kOutsideCode) = u32x2{outside, outside};
    }

    // ---- one item: tile `tile_in_launch` of the launch, all of its rows or (cut != 0) its row block `block` ---------------
    auto run_item = [&](const uint32_t tile_arg, const uint32_t block_arg /* kWholeTile: all rows */,
                        const uint32_t slot_arg /* cut tiles: the tile's hand-off slot */) -> bool {   // false: stop (abort requested, or a hand-off never came)
        // (wave-uniform, but a division may have left them in vector registers)
        const uint32_t tile_in_launch = __builtin_amdgcn_readfirstlane(tile_arg), block = __builtin_amdgcn_readfirstlane(block_arg);
        const uint32_t handoff_slot = __builtin_amdgcn_readfirstlane(slot_arg);
        const uint32_t cut = block != kWholeTile ? 1u : 0u;
        const rare_args_t launch = rare_args();
        const uint32_t tile = launch->tile_begin + tile_in_launch;
        const int64_t d0 = launch->first_diag + (int64_t)tile * kTileDiags;     // wave's first diagonal
        // steps whose cells of this tile can lie inside the matrix
        // (rows fit 24 bits: once clamped to [0, nrows_padded] everything about rows is 32-bit scalar arithmetic)
        int64_t lo64 = -d0 - kTileDiags;                                // first chunk touching column >= 0
        if (lo64 < 0) lo64 = 0;
        if (lo64 > (int64_t)nrows_padded) lo64 = nrows_padded;
        int64_t hi64 = launch->col_end - d0;                                  // first chunk entirely right of the shard's columns
        if (hi64 > (int64_t)nrows_padded) hi64 = nrows_padded;
        if (hi64 < 0) hi64 = 0;
        const uint32_t p_lo = __builtin_amdgcn_readfirstlane((uint32_t)lo64), p_hi = __builtin_amdgcn_readfirstlane((uint32_t)hi64);
        uint32_t p_begin = p_lo, p_end = p_hi;                          // this item's share of them
        if (cut) {
            // the block's rows [b0, b1): from the table for the first blocks, uniform blocks behind them
            const uint32_t ncuts = launch->ncuts, uniform_rows = launch->uniform_rows, last_cut = launch->row_cut[ncuts];
            const uint32_t b0 = block < ncuts ? launch->row_cut[block] : last_cut + (block - ncuts) * uniform_rows;
            const uint32_t b1 = block < ncuts ? launch->row_cut[block + 1] : last_cut + (block - ncuts + 1) * uniform_rows;
            if (p_begin < b0) p_begin = b0;
            if (p_end > b1) p_end = b1;
        }
        if (p_begin >= p_end) return true;
        // An abort request stops every item that has not started yet: a run of short models, whose items never reach the
        // 2048-row poll below, drains at once too.  The word is read past the caches -- a trip to memory -- so the load is
        // ISSUED here and looked at behind the item's first loads (symbols, model rows), with which it travels: tested at
        // once, it cost a one-chunk tile a memory latency of its own before anything else had started.
        uint32_t abort_now = 0;
        {
            const uint32_t* const abort_flag = rare_args()->abort_flag;
            if (abort_flag) abort_now = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }

        uint32_t x[kRegs], x2[kRegs];     // the scores and their ping-pong partner (see step_window)
        // (uniform base, lane offset added where it is used: a per-lane 64-bit pointer would sit in two VGPRs through the item)
        if (p_begin > p_lo) {
            // the rows above belong to the previous row block of this tile: wait for it, take over its scores
            const rare_args_t rare = rare_args();
            const uint32_t* const abort_flag = rare->abort_flag;
            uint32_t* const block_flags = rare->block_flags;
            uint32_t* const fault = rare->fault;
            const uint32_t* const tile_state = rare->block_state + (size_t)handoff_slot * (kRegs / 2 * 64);
            uint32_t spins = 0;
            while (__hip_atomic_load(block_flags + handoff_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < block) {
                __builtin_amdgcn_s_sleep(32);
                ++spins;
                if ((spins & 63u) == 0) {
                    // the block before may never come: the run was aborted (its wave left without a hand-off), or a
                    // wave somewhere gave up waiting
                    if (abort_flag && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
                    if (__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
                }
                if (spins == kHandoffSpins) {
                    if (fresh_lane() == 0) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t my_lane = fresh_lane();
            // (between chunks every cell is a multiple of 256 -- no mark is pending -- so a cell travels as its high byte: 2 KB per hand-off)
#pragma unroll
            for (int i = 0; i < kRegs / 2; i++) {
                const uint32_t packed = tile_state[i * 64 + my_lane];
                x[2 * i] = __builtin_amdgcn_perm(0u, packed, 0x010c000cu);
                x[2 * i + 1] = __builtin_amdgcn_perm(0u, packed, 0x030c020cu);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kRegs; i++) x[i] = kScoreZero;
        }
#pragma unroll
        for (int i = 0; i < kRegs; i++) x2[i] = kScoreZero;

        // Window of the current chunk, symbol positions [j, j+64) with j = d0 + 32 lane + p0 (see LazySymbols)
        uint32_t C[32];
        // The 32 symbols (8 bytes) of this lane at position d0 + rel + 32 lane, rel wave-uniform, ready for expansion;
        // the wave's 512 bytes are consecutive: a uniform base and the lane's byte offset.
        // In three parts (the scalars and the loaded words travel separately: in one struct, the lane-dependent branch around the
        // loads made hipcc treat the scalars as per-lane values too.)
        struct SymbolRange { int64_t first; int32_t valid_lo, valid_hi; uint32_t edge; };
        auto symbol_range = [&](int64_t rel) -> SymbolRange {
            SymbolRange e;
            e.first = d0 + rel;                                           // the wave's first position
            // positions relative to `first` that lie inside [0, N), clamped to [0, 4096]: all of it on the scalar unit
            e.valid_lo = clamp_to_4096(-e.first);
            e.valid_hi = clamp_to_4096(nsymbols - e.first);
            e.edge = (uint32_t)(e.valid_lo != 0) | (uint32_t)(e.valid_hi < kTileDiags);   // first < 0, or first + 2048 > N
            return e;
        };
        auto load_symbols = [&](const SymbolRange& e, const uint32_t lane8, uint2& w, uint32_t& separators) {
            const uint8_t* const base = seq + (e.first >> 2);           // only dereferenced for lanes inside [0, N)
            w = make_uint2(0u, 0u);
            separators = 0;
            if (!e.edge) {                                                // the usual case: one coalesced load, uniform base + lane offset
                // (lane8 comes fresh from LDS in this block: its zero-extension folds into the load's "SGPR base + 32-bit VGPR offset" form)
                w = *reinterpret_cast<const uint2*>(base + lane8);
                if (fresh_uniform(has_mask)) separators = rare_args()->pair_mask[(e.first >> 5) + (lane8 >> 3)];
            } else if ((int32_t)(lane8 * 4) >= e.valid_lo && (int32_t)(lane8 * 4) + 32 <= e.valid_hi) {
                w = *reinterpret_cast<const uint2*>(base + lane8);
                if (fresh_uniform(has_mask)) separators = rare_args()->pair_mask[(e.first >> 5) + (lane8 >> 3)];
            }
        };
        auto finish_symbols = [&](const SymbolRange& e, const uint2 w, const uint32_t separators, LazySymbols& z) {
            z.valid_lo = e.valid_lo; z.valid_hi = e.valid_hi;
            z.table_base = table_base;
            z.separators = separators;
            uint32_t special = e.edge;                                    // from scalars and a ballot only: stays on the scalar unit
            if (fresh_uniform(has_mask)) special |= __any(z.separators != 0) ? 1u : 0u;
            z.special = opaque_uniform(special ? kFlagSpecial : 0u);
            prepare_symbols(z, w.x, w.y);
        };
        auto fetch_symbols = [&](int64_t rel, LazySymbols& z, const uint32_t lane8) {
            const SymbolRange e = symbol_range(rel);
            uint2 w; uint32_t separators;
            load_symbols(e, lane8, w, separators);
            finish_symbols(e, w, separators, z);
        };
        // A chunk's 16 step-pair tables: entry (a,b) of pair P = { word(step 2P, a), word(step 2P+1, b) } with
        // word(t, a) = (M[t][a] << 8) | (M[t-1][a] << 24); rows[] is shifted by one, so rows[t] is M[t-1].
        // Four lanes share a step pair: they fetch its three model rows and each writes the entries of one b.
        struct ModelRows { uint32_t r0, r1, r2, sel_second, entries; };
        auto fetch_rows = [&](uint32_t p0, const LaneWords& mine) -> ModelRows {
            const uint32_t* r = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(rows + p0) + mine.row_offset);
            return ModelRows{r[0], r[1], r[2], mine.sel_second, mine.entries};
        };
        auto build_tables = [&](const ModelRows r) {
            const uint32_t second = __builtin_amdgcn_perm(r.r1, r.r2, r.sel_second);
            const lds_words_out_t out = (lds_words_out_t)(uintptr_t)(table_base + r.entries);
#pragma unroll
            for (int a = 0; a < 4; a++) out[a] = u32x2{__builtin_amdgcn_perm(r.r0, r.r1, word_selector(a)), second};
        };
        LazySymbols z;
        LaneWords mine = read_lane_words(lane_words_address);
        fetch_symbols(p_begin, z, mine.lane8);
        uint32_t lower_special = z.special & kFlagSpecial;                 // (wave-uniform; see the chunk loop)
        if (__builtin_amdgcn_readfirstlane(abort_now)) return false;       // (a scalar branch: every lane read the same word)
        expand_all(C, z, std::make_integer_sequence<int, 16>{});       // -> C[16..32): the first chunk's lower half after the slide
        ModelRows next_rows = fetch_rows(p_begin, mine);
        fetch_symbols(p_begin + 32, z, mine.lane8);
        // The global loads of a chunk (12 B of model rows and 8 B of symbols per lane) are issued at the END of the chunk
        // before it and consumed at its top: no register carries them across the windows, and the latency that is
        // exposed this way is covered by the other five waves of the SIMD.  Round 4 checked that it really is: an ablation
        // without the two loads and the table build (wrong results) runs 4.4 % faster at 1024 rows, but that is the WORK that
        // went with them -- with both waits truly removed (the model's rows kept in LDS by the workgroup, the next chunk's
        // symbols loaded in front of window 5 into registers that are dead by then; correct, 78 GPU tests) the kernel is 1.3 %
        // SLOWER (1.768-1.774 against 1.747 ms: 633 instead of 627 vector instructions per chunk and four spilled registers),
        // either one alone changes nothing or costs 1 %, and all five words staged through LDS by LDS-DMA a chunk ahead
        // cost 2 % (an LDS-DMA instruction costs the wave more issue time than the wait it saves).  DESIGN.md section 7b.

        bool go_on = true;                 // wave-uniform
        // one bit per chunk: may the chunk look for hits every four steps only (ssv_prepare_model)?
        // The word of the current 1024 rows and, loaded a whole block ahead, the next one: the test never waits for memory.
        // With a separator mask (round 5; before: never) a chunk may, too, unless a separator pair or the matrix's edge lies in the wave's
        // window of 64 symbols: the "outside" entry then scores -128, two of them take a crossed cell's mark away inside a window.
        // `lower_special`: the flag of the symbols that have slid into the window's lower half (z.special holds the upper half's).
        uint32_t safe_now = safe_chunks[p_begin >> 7];             // (a byte per chunk: a word holds 128 rows' worth)
        uint32_t safe_next = safe_chunks[(p_begin >> 7) + 1];
        for (uint32_t p0 = p_begin; p0 < p_end; p0 += kChunkRows) {
            if ((p0 & 127) == 0 && p0 != p_begin) {
                safe_now = safe_next;
                safe_next = safe_chunks[(p0 >> 7) + 1];
            }
            // abort: a device word, read past the caches every 2048 rows inside an item (and between items, below)
            if (((p0 & 2047) == 0) && p0 != p_begin) {
                const uint32_t* const abort_flag = rare_args()->abort_flag;
                if (abort_flag && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { go_on = false; break; }
            }
            build_tables(next_rows);
            // slide the window by 32 symbols
#pragma unroll
            for (int k = 0; k < 16; k++) C[k] = C[k + 16];
            {
                // (bit arithmetic on 0 / 1 values, no select: a select between uniform values went through v_cndmask + v_readfirstlane here)
                static_assert(kFlagSpecial == 2u, "the flag's position is used below");
                const uint32_t upper_special = z.special & kFlagSpecial;
                const uint32_t blocked = fresh_uniform(has_mask) & ((lower_special | upper_special) >> 1);      // 1: a separator or the edge in the window, under a mask
                const uint32_t safe_bits = ((safe_now >> ((p0 >> 2) & 24u)) & 0xffu) & (blocked - 1u);         // the chunk's byte, or nothing
                lower_special = upper_special;                    // (what it is once this chunk's symbols have slid down)
                z.special = opaque_uniform((safe_bits << kFlagSafeShift) | upper_special);     // + the windows' flags for this chunk
            }
            step_windows<Trace, false>(x, x2, C, z, z.special, sink, staged, p0, d0, std::make_integer_sequence<int, kChunkRows / kWindowSteps>{});
            asm volatile("" ::: "memory");                        // keeps hipcc from hoisting the loads above the windows
            mine = read_lane_words(lane_words_address);
            next_rows = fetch_rows(p0 + kChunkRows, mine);        // rows[] has kModelSlack words behind the model
            fetch_symbols(p0 + kChunkRows + 32, z, mine.lane8);
        }
        if (!go_on) return false;
        if (p_end == nrows_padded) {
            // the high cells run one row behind: one more step gives them the model's last row
            build_tables(next_rows);                               // fetched for p_end by the last chunk
#pragma unroll
            for (int k = 0; k < 16; k++) C[k] = C[k + 16];
            step_last<Trace, false, 0>(x, x2, C, sink, staged, p_end, d0, table_base, std::make_integer_sequence<int, kRegs>{});
        } else if (p_end < p_hi) {
            // the tile goes on in the next row block: hand the scores over (release: the stores, then the count)
            const rare_args_t rare = rare_args();
            uint32_t* const tile_state = rare->block_state + (size_t)handoff_slot * (kRegs / 2 * 64);
            const uint32_t my_lane = fresh_lane();
#pragma unroll
            for (int i = 0; i < kRegs / 2; i++) tile_state[i * 64 + my_lane] = __builtin_amdgcn_perm(x[2 * i + 1], x[2 * i], 0x07050301u);   // the four high bytes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (my_lane == 0) __hip_atomic_store(rare->block_flags + handoff_slot, block + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    };

    // ---- items --------------------------------------------------------------------------------------------------------
    // One item per wave, four per block, the blocks handed out by the hardware's block scheduler.  Block i belongs to
    // partition i mod 8 -- blocks are dealt round-robin over the eight XCDs, so a partition's blocks share an L2 (observed
    // placement, used for speed only: nothing below depends on it) -- and a partition owns a RUN of adjacent units (runs of
    // equal work, not of equal length: the tiles at the matrix's two ends cover fewer rows): the
    // waves that stream the same stretch of the sequence, 2048 columns apart, run on one XCD at about the same time, and a
    // row block is handed over inside one L2.  A partition's items: its whole units first, then the row blocks of its last
    // split_units units, row-block-major.  Launches without cut tiles: the block's number says which four items it takes.
    // With cut tiles the BLOCK draws a ticket from its partition's counter when it starts (one returning atomic per four
    // items): the order of a partition's items is then the order in which its blocks really started, whatever order the
    // hardware dispatched them in, and the row block a wave waits for (a smaller item number of the same partition) is
    // held by a block that has started.
    {
        const rare_args_t launch = rare_args();
        const uint32_t parts_log2 = launch->parts_log2, split_units = launch->split_units;
        const uint32_t part = blockIdx.x & ((1u << parts_log2) - 1u);
        uint32_t ticket = blockIdx.x >> parts_log2;
        if (split_units) {
            __shared__ uint32_t block_ticket;
            if (wave == 0 && fresh_lane() == 0)
                block_ticket = __hip_atomic_fetch_add(launch->tickets + part * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            ticket = __builtin_amdgcn_readfirstlane(block_ticket);
        }
        const uint32_t item = ticket * kWavesPerBlock + wave;         // wave-uniform; within the partition
        // A partition's tiles [t0, t1), in the order its items take them: groups of tiles_per_item adjacent tiles (a wave WALKS a
        // group; experiments on short models only), single tiles, then the row blocks of its last split_units tiles,
        // row-block-major.  (Decoded afresh for every tile of a walk -- a dozen scalar instructions: only `item` and the count
        // stay alive across a tile, and the hot loop is as short of scalar registers as of vector ones.  ONE call site of run_item:
        // with two, hipcc stops inlining it -- and a called function cannot read the kernarg segment.)
        for (uint32_t g = 0;; g++) {
            const rare_args_t plan = rare_args();
            const uint32_t t0 = plan->part_begin[part], t1 = plan->part_begin[part + 1u];
            const uint32_t mine = t1 - t0, cut_limit = plan->split_units;
            const uint32_t cut_units = cut_limit < mine ? cut_limit : mine, whole = mine - cut_units;
            const uint32_t per_group = plan->tiles_per_item, single_tiles = plan->single_tiles;
            const uint32_t groups = per_group > 1 ? (whole - (single_tiles < whole ? single_tiles : whole)) / per_group : 0u;
            const uint32_t singles = whole - groups * per_group;
            // A cut tile's hand-off slot (SsvRare::block_flags / block_state): cut tile u of partition k has slot k * split_units + u
            // -- the buffers hold the launch's CUT tiles only (C4's rank: 9,216 of 61,000 tiles; a 3 Gbp genome against a tall
            // model: 18 MB instead of 3 GB) -- or, where every tile is cut (split_units = all ones), the tile's own number.
            uint32_t first_tile = 0, walk = 0, block = kWholeTile, slot = 0;
            if (item < groups) {
                first_tile = t0 + item * per_group;
                walk = per_group;
            } else if (item - groups < singles) {
                first_tile = t0 + groups * per_group + (item - groups);
                walk = 1;
            } else if (item - groups - singles < cut_units * plan->nrow_blocks) {
                const uint32_t j = item - groups - singles;
                block = j / cut_units;
                const uint32_t unit = j - block * cut_units;
                first_tile = t0 + whole + unit;
                slot = cut_limit == 0xffffffffu ? first_tile : part * cut_limit + unit;
                walk = 1;
            }
            if (g >= walk || !run_item(first_tile + g, block, slot)) break;
        }
    }

    leave_block(lds->stage, staged, wave);
}

// ---- the resident-table variant (short models) ---------------------------------------------------------------------------
// A tile of a short model -- one to eight 32-row chunks -- is mostly prologue in the kernel above: of the ~22,500 cycles a
// one-chunk tile's wave lives, the chunk takes 11,900; the rest is latency in a row (a workgroup's start, the item's decode,
// the first symbols, then window + second symbols + model rows, the rows of the step behind the chunk) and ~2 us of empty
// wave slot between a workgroup's end and its successor's start.  The reference's array has no such
// dependence on the model's height (README.md:4; device/HavacHls.cpp:220-319: one row per clock whatever L is).  What a
// short model makes possible:
//   * ALL its tables fit LDS at once -- (chunks + 1) x 2.25 KB, the same for every tile of the launch -- so a workgroup builds
//     them ONCE (its four waves share them) instead of once per chunk and wave: no model rows loaded, no v_perm, no table
//     write, no wait for either inside a tile.  A window entry is `table | code*8`; the entries that slide from one chunk's
//     window into the next use the next table through the read's offset field (slid_offset), so nothing is re-based;
//   * the launch is as many workgroups as the chip holds at once, and every wave WALKS its own run of adjacent tiles (equal
//     runs, handed out by the wave's number alone): a workgroup's start, end and empty slot are paid once per launch, not
//     once per four tiles, and nothing but the wave's own stage flush ever touches the hit counter;
//   * the next tile's first symbols are fetched before the step behind the current tile's last chunk.
// No row blocks, no trace: the host (havac_dev.hip, pick_resident_kernel) takes the standard kernel there.  A separator mask: the
// second instantiation, ssv_resident_kernel_masked (round 5).
#ifdef HAVAC_WAVE_CLOCKS
// experiments only (tools/wave_clocks.py): when each wave of the resident-table kernel started its run and left it, on the
// chip-wide 100 MHz clock, and which SIMD it ran on
__device__ uint64_t g_wave_clocks[4 * 16384];
#endif
constexpr uint32_t kResidentRows = 256;                             // models of up to this many (padded) rows
constexpr int kResidentTables = kResidentRows / kChunkRows + 1;     // + the table of the step behind the last chunk
struct __attribute__((aligned(128))) ResidentLds {
    uint64_t stage[kWavesPerBlock][kHitStage];      // in front of the tables: "one table below table 0" (the base the first
    uint8_t table[kResidentTables][kTableBytes];    //   chunk's slid entries carry) must still be an LDS address
};
static_assert(sizeof(uint64_t) * kWavesPerBlock * kHitStage >= kTableBytes, "the stages cover one table's worth of addresses");

struct TileSymbols { uint2 w0, w1; uint32_t abort_word; uint32_t separators; };      // what a tile's prologue needs from memory (separators, masked launches only: the 16 separator bits of w0, and of w1 << 16)

template <bool Masked /* a separator mask is set (SsvRare::pair_mask): the second instantiation, ssv_resident_kernel_masked */>
__device__ __forceinline__ void ssv_resident_body(const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                                                  const uint32_t* __restrict__ safe_chunks, const int64_t nsymbols,
                                                  const uint32_t nrows_padded) {
    __shared__ ResidentLds lds;
#ifdef HAVAC_SLOW_CLOCKS
    if (threadIdx.x == 0) { s_slow_clocks[0] = s_slow_clocks[1] = 0; s_slow_clocks[2] = __builtin_readcyclecounter(); s_slow_clocks[3] = wall_clock64(); }
#endif
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const HitSink sink{lds.stage[wave]};
    uint32_t staged = 0;          // wave-uniform
    const uint32_t tables = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds.table);
    const uint32_t nchunks = nrows_padded / kChunkRows;

    // ---- the tables, once per workgroup: table c = chunk c's 16 step-pair tables (see build_tables in ssv_diag_body), table
    // nchunks = the step behind the last chunk (its pair 0: the model's last row for the high cells, a padding row for the low)
    {
        const uint32_t lane = fresh_lane();
        const uint32_t my_pair = lane >> 2, my_b = lane & 3;
        for (uint32_t c = wave; c <= nchunks; c += kWavesPerBlock) {
            const uint32_t* const r = rows + c * kChunkRows + my_pair * 2;       // rows[] is shifted by one: rows[t] is M[t-1]
            const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
            const uint32_t second = __builtin_amdgcn_perm(r1, r2, word_selector(my_b));
            const uint32_t at = tables + c * kTableBytes + my_pair * kPairStride;
            const lds_words_out_t out = (lds_words_out_t)(uintptr_t)(at + my_b * 32);
#pragma unroll
            for (int a = 0; a < 4; a++) out[a] = u32x2{__builtin_amdgcn_perm(r0, r1, word_selector(a)), second};
            // (with a separator mask the "outside" entries score -128, as in ssv_diag_body)
            if (my_b == 0) *(lds_words_out_t)(uintptr_t)(at + kOutsideCode) = u32x2{Masked ? kOutsideReset : kOutsideNeutral, Masked ? kOutsideReset : kOutsideNeutral};
        }
    }
    __syncthreads();

    // ---- this wave's run of tiles
    uint32_t tile, tile_end;
    {
        const rare_args_t plan = rare_args();
        const uint32_t g = blockIdx.x * kWavesPerBlock + wave, slots = plan->walk_slots, last = plan->walk_rounds - 1u;
        uint32_t round = g / slots;
        if (round > last) round = last;
        const uint32_t len = plan->walk_len[round];
        tile = plan->walk_base[round] + (g - round * slots) * len;
        tile_end = tile + len;
        const uint32_t ntiles = plan->ntiles;
        if (tile > ntiles) tile = ntiles;
        if (tile_end > ntiles) tile_end = ntiles;
    }
    // Round 5: where the model's rows end in the first half of its last chunk (rows mod 32 in 1 .. 16) the chunk's second half -- sixteen
    // steps of padding rows that change no score -- is not run: a 40-row model costs 48 steps, not 64
    const uint32_t half_tail = opaque_uniform(rare_args()->half_tail);
    // (a byte per chunk: a model of up to 256 rows lies inside the first two flag words)
    const uint64_t safe_words = (uint64_t)safe_chunks[0] | ((uint64_t)safe_chunks[1] << 32);

    struct SymbolRange { int64_t first; int32_t valid_lo, valid_hi; uint32_t edge; };
    auto symbol_range = [&](int64_t first) -> SymbolRange {          // the wave's 2048 positions from `first` on (see ssv_diag_body)
        SymbolRange e;
        e.first = first;
        e.valid_lo = clamp_to_4096(-first);
        e.valid_hi = clamp_to_4096(nsymbols - first);
        e.edge = (uint32_t)(e.valid_lo != 0) | (uint32_t)(e.valid_hi < kTileDiags);
        return e;
    };
    auto load_symbols = [&](const SymbolRange& e, const uint32_t lane8, uint32_t& separators) -> uint2 {
        const uint8_t* const base = seq + (e.first >> 2);             // only dereferenced for lanes inside [0, N)
        uint2 w = make_uint2(0u, 0u);
        separators = 0;
        if (!e.edge || ((int32_t)(lane8 * 4) >= e.valid_lo && (int32_t)(lane8 * 4) + 32 <= e.valid_hi)) {
            w = *reinterpret_cast<const uint2*>(base + lane8);
            if constexpr (Masked) separators = rare_args()->pair_mask[(e.first >> 5) + (lane8 >> 3)];
        }
        return w;
    };
    auto finish_symbols = [&](const SymbolRange& e, const uint2 w, const uint32_t separators, const uint32_t table_base, LazySymbols& z) {
        z.valid_lo = e.valid_lo; z.valid_hi = e.valid_hi;
        z.table_base = table_base;
        z.separators = Masked ? separators : 0u;
        uint32_t special = e.edge;
        if constexpr (Masked) special |= __any(separators != 0) ? 1u : 0u;
        z.special = opaque_uniform(special ? kFlagSpecial : 0u);
        prepare_symbols(z, w.x, w.y);
    };
    // a tile's first diagonal and the first of its rows that can lie inside the matrix (ssv_diag_body: d0, p_lo)
    // (the launch's tiling is read ONCE: this kernel has scalar registers to spare -- 77 -- and a tile of one chunk has no time
    // for three kernarg loads in a row)
    const int64_t tiling_diag0 = rare_args()->first_diag + (int64_t)rare_args()->tile_begin * kTileDiags, tiling_col_end = rare_args()->col_end;
    auto tile_start = [&](const uint32_t t, int64_t& d0, uint32_t& p_lo) {
        d0 = tiling_diag0 + (int64_t)t * kTileDiags;
        int64_t lo64 = -d0 - kTileDiags;
        if (lo64 < 0) lo64 = 0;
        if (lo64 > (int64_t)nrows_padded) lo64 = nrows_padded;
        p_lo = __builtin_amdgcn_readfirstlane((uint32_t)lo64);
    };
    // A tile's first loads, all in flight before anything waits: the symbols of its first two chunks and, for a run's first tile,
    // the abort word (read past every cache: a trip of microseconds, as long as a whole one-chunk tile -- once per run, which
    // is at most a few dozen microseconds long, is often enough).  The ranges are recomputed where the words are consumed --
    // scalar arithmetic -- so that only the loaded words are carried.
    auto issue_tile_loads = [&](const uint32_t t, TileSymbols& L, const bool with_abort_word) {
        int64_t t_d0; uint32_t t_lo;
        tile_start(t, t_d0, t_lo);
        const uint32_t lane8 = fresh_lane() * 8u;
        uint32_t s0, s1;
        L.w0 = load_symbols(symbol_range(t_d0 + t_lo), lane8, s0);
        L.w1 = load_symbols(symbol_range(t_d0 + t_lo + 32), lane8, s1);
        L.separators = Masked ? (s0 | (s1 << 16)) : 0u;
        L.abort_word = 0;
        if (with_abort_word) {
            const uint32_t* const abort_flag = rare_args()->abort_flag;
            L.abort_word = abort_flag ? __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
        }
    };

    TileSymbols ahead{};               // the first loads of the run's NEXT tile, in flight across the end of the current one
    auto run_tile = [&](const uint32_t tile_arg, const uint32_t more_arg /* 1: another tile follows in this wave's run */) -> bool {
        const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_arg), more = __builtin_amdgcn_readfirstlane(more_arg);
        int64_t d0; uint32_t p_lo;
        tile_start(tile, d0, p_lo);
        int64_t hi64 = tiling_col_end - d0;                                        // first row entirely right of the shard's columns
        if (hi64 > (int64_t)nrows_padded) hi64 = nrows_padded;
        if (hi64 < 0) hi64 = 0;
        const uint32_t p_hi = __builtin_amdgcn_readfirstlane((uint32_t)hi64);
        const TileSymbols now = ahead;
        // (`ahead` is written on every path: kept conditionally, its OLD words would stay alive through the chunk loop)
        ahead = TileSymbols{};
        if (p_lo >= p_hi) {                                              // (a tile without rows: never expected inside a launch's range)
            if (more) issue_tile_loads(tile + 1, ahead, false);
            return true;
        }
        if (__builtin_amdgcn_readfirstlane(now.abort_word)) return false;       // abort: a device word, looked at once per run
        uint32_t x[kRegs], x2[kRegs];     // the scores and their ping-pong partner (see step_window)
#pragma unroll
        for (int i = 0; i < kRegs; i++) x[i] = x2[i] = kScoreZero;
        uint32_t C[32];
        LazySymbols z;
        const uint32_t first_table = tables + (p_lo / kChunkRows) * kTableBytes;
        finish_symbols(symbol_range(d0 + p_lo), now.w0, now.separators & 0xffffu, first_table - kTableBytes, z);
        uint32_t lower_special = z.special & kFlagSpecial;             // (masked launches: see ssv_diag_body's chunk loop)
        expand_all(C, z, std::make_integer_sequence<int, 16>{});       // -> C[16..32): the first chunk's lower half after the slide
        finish_symbols(symbol_range(d0 + p_lo + 32), now.w1, now.separators >> 16, first_table, z);
        for (uint32_t p0 = p_lo; p0 < p_hi; p0 += kChunkRows) {
#pragma unroll
            for (int k = 0; k < 16; k++) C[k] = C[k + 16];
            if constexpr (Masked) {
                // a chunk may test every four steps unless a separator pair or the matrix's edge lies in the wave's window
                const uint32_t upper_special = z.special & kFlagSpecial;
                const uint32_t blocked = (lower_special | upper_special) >> 1;      // (bit arithmetic, no select: see ssv_diag_body)
                const uint32_t safe_bits = ((uint32_t)(safe_words >> (p0 >> 2)) & 0xffu) & (blocked - 1u);
                lower_special = upper_special;
                z.special = opaque_uniform((safe_bits << kFlagSafeShift) | upper_special);
            } else {
                z.special = opaque_uniform((((uint32_t)(safe_words >> (p0 >> 2)) & 0xffu) << kFlagSafeShift) | (z.special & kFlagSpecial));      // + the windows' flags for this chunk
            }
            step_windows<false, true>(x, x2, C, z, z.special, sink, staged, p0, d0, std::integer_sequence<int, 0, 1, 2, 3>{});
            if (!(fresh_uniform(half_tail) && p0 + kChunkRows == nrows_padded))      // (not where the model's rows end in this chunk's first half)
                step_windows<false, true>(x, x2, C, z, z.special, sink, staged, p0, d0, std::integer_sequence<int, 4, 5, 6, 7>{});
            asm volatile("" ::: "memory");                        // keeps hipcc from hoisting the loads above the windows
            if (p0 + kChunkRows < p_hi) {
                // the next chunk's upper half (issued at the end of a chunk, consumed at the top of the next: see ssv_diag_body)
                const SymbolRange e = symbol_range(d0 + p0 + kChunkRows + 32);
                uint32_t separators;
                const uint2 w = load_symbols(e, fresh_lane() * 8u, separators);
                finish_symbols(e, w, separators, tables + ((p0 >> 5) + 1u) * kTableBytes, z);
            }
        }
        // the next tile's first loads: in flight across the step behind the last chunk and the next tile's scalar prologue
        // (issued here, not inside the chunk loop: nothing of them is alive through the windows)
        if (more) issue_tile_loads(tile + 1, ahead, false);
        if (p_hi == nrows_padded) {
            if (fresh_uniform(half_tail)) {
                // the step behind the model's last row lies in the middle of the last chunk: its pair 8, no slide
                expand_entry<7>(C, z);      // (entry 23: what window 4 would have expanded)
                step_last<false, true, 4>(x, x2, C, sink, staged, p_hi - kChunkRows, d0, 0u, std::make_integer_sequence<int, kRegs>{});
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) C[k] = C[k + 16];
                step_last<false, true, 0>(x, x2, C, sink, staged, p_hi, d0, 0u, std::make_integer_sequence<int, kRegs>{});
            }
        }
        return true;
    };
#ifdef HAVAC_WAVE_CLOCKS
    const uint64_t clock_start = __builtin_readcyclecounter() * 0 + wall_clock64();
    const uint32_t tiles_mine = tile_end - tile;
#endif
    if (tile < tile_end) issue_tile_loads(tile, ahead, true);
    for (; tile < tile_end; tile++) {
        if (!run_tile(tile, tile + 1 < tile_end ? 1u : 0u)) break;
    }
#ifdef HAVAC_WAVE_CLOCKS
    {
        const uint32_t g = blockIdx.x * kWavesPerBlock + wave;
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        if (fresh_lane() == 0 && g < 16384u) {
            g_wave_clocks[4 * g] = clock_start; g_wave_clocks[4 * g + 1] = wall_clock64();
            g_wave_clocks[4 * g + 2] = hw_id; g_wave_clocks[4 * g + 3] = tiles_mine;
        }
    }
#endif
    leave_block(lds.stage[wave], staged, wave);
}

// The tails of `nblocks` blocks (tail_counts[b] sort keys at tails[b * kTailSlots ...]) are appended to the queue.  A
// workgroup takes kGatherBlocks tails: one lane per tail sums the counts up (a scan over a wave), ONE returning atomic
// reserves the workgroup's stretch of the queue, and each of the four waves copies its share of the tails with all its
// loads in flight before the first store (a tail is at most two keys per lane).
constexpr int kGatherBlocks = 32;
__global__ __launch_bounds__(256)
void ssv_gather_tails(const uint64_t* __restrict__ tails, const uint32_t* __restrict__ tail_counts, uint32_t nblocks,
                      uint64_t* __restrict__ hits, unsigned long long* hit_count, uint64_t capacity) {
    static_assert(kTailSlots == 128 && kGatherBlocks == 32, "two keys per lane and tail, eight tails per wave");
    __shared__ uint32_t count[kGatherBlocks], offset[kGatherBlocks];
    __shared__ unsigned long long base;
    const uint32_t first = blockIdx.x * kGatherBlocks;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        const uint32_t n = (lane < (uint32_t)kGatherBlocks && first + lane < nblocks) ? tail_counts[first + lane] : 0u;
        uint32_t inclusive = n;
#pragma unroll
        for (int d = 1; d < kGatherBlocks; d *= 2) {
            const uint32_t up = __shfl_up(inclusive, d, 64);
            if ((int)lane >= d) inclusive += up;
        }
        if (lane < (uint32_t)kGatherBlocks) { count[lane] = n; offset[lane] = inclusive - n; }
        if (lane == kGatherBlocks - 1) base = inclusive ? atomicAdd(hit_count, (unsigned long long)inclusive) : 0ull;
    }
    __syncthreads();
    constexpr int kPerWave = kGatherBlocks / 4;
    uint64_t key[kPerWave][2];
#pragma unroll
    for (int k = 0; k < kPerWave; k++) {
        const uint32_t r = wave * kPerWave + k;
        const uint64_t* const src = tails + (size_t)(first + r) * kTailSlots;
        const uint32_t n = count[r];
        key[k][0] = lane < n ? src[lane] : 0;
        key[k][1] = lane + 64 < n ? src[lane + 64] : 0;
    }
#pragma unroll
    for (int k = 0; k < kPerWave; k++) {
        const uint32_t r = wave * kPerWave + k;
        const uint32_t n = count[r];
        const unsigned long long dst = base + offset[r];
        if (lane < n && dst + lane < capacity) hits[dst + lane] = key[k][0];
        if (lane + 64 < n && dst + lane + 64 < capacity) hits[dst + lane + 64] = key[k][1];
    }
}

__global__ __launch_bounds__(64 * kWavesPerBlock, HAVAC_WAVES_PER_SIMD)
void ssv_diag_kernel(const SsvRare /* read through rare_args(), never by name */,
                     const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                     const uint32_t* __restrict__ safe_chunks /* one bit per 32-row chunk: a hit test every four steps is exact there (with a separator mask: where no separator lies in the window) */,
                     const int64_t nsymbols, const uint32_t nrows_padded) {
    ssv_diag_body<false>(seq, rows, safe_chunks, nsymbols, nrows_padded);
}

// short models (see "the resident-table variant"): as many workgroups as the chip holds, every wave walks its run of tiles
#ifndef HAVAC_RESIDENT_WAVES
#define HAVAC_RESIDENT_WAVES HAVAC_WAVES_PER_SIMD
#endif
__global__ __launch_bounds__(64 * kWavesPerBlock, HAVAC_RESIDENT_WAVES)
void ssv_resident_kernel(const SsvRare, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                         const uint32_t* __restrict__ safe_chunks, const int64_t nsymbols, const uint32_t nrows_padded) {
    ssv_resident_body<false>(seq, rows, safe_chunks, nsymbols, nrows_padded);
}
// the same with a separator mask (round 5: a short model against a FASTA of several records -- the file-level API's usual case --
// no longer falls back to the standard kernel).  An instantiation of its own: the separator word of a tile's first symbols is one
// more word alive across the end of the tile before, which at 79 of 80 registers costs 16 B of scratch per lane -- written and read
// once per TILE, outside the windows -- that launches without a mask do not pay.
__global__ __launch_bounds__(64 * kWavesPerBlock, HAVAC_RESIDENT_WAVES)
void ssv_resident_kernel_masked(const SsvRare, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                                const uint32_t* __restrict__ safe_chunks, const int64_t nsymbols, const uint32_t nrows_padded) {
    ssv_resident_body<true>(seq, rows, safe_chunks, nsymbols, nrows_padded);
}

// the same body with the per-cell trace compiled in (see CellRecord): a debugging aid, never launched unless a trace window is set
__global__ __launch_bounds__(64 * kWavesPerBlock)
void ssv_diag_kernel_traced(const SsvRare, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ rows,
                            const uint32_t* __restrict__ safe_chunks, const int64_t nsymbols, const uint32_t nrows_padded) {
    ssv_diag_body<true>(seq, rows, safe_chunks, nsymbols, nrows_padded);
}

// ---------------------------------------------------------------------------
// after the radix sort of the keys: rewrite them as the reference's records
// (grid-stride: a launch holds fewer than 2^32 threads -- the dispatch packet counts work-items in 32 bits -- and C4 on
// one GPU orders 4.5e9 records)
__global__ void ssv_keys_to_records(uint64_t* __restrict__ hits, uint64_t n, uint32_t row_bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        hits[i] = key_to_record(hits[i], row_bits);
}
__global__ void ssv_records_to_keys(uint64_t* __restrict__ hits, uint64_t n, uint32_t row_bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        hits[i] = record_to_key(hits[i], row_bits);
}

}  // namespace havac
