// havac_dev.hip -- C ABI of the MI355X device layer (include/havac_dev.h).
//
// Replaces the reference's XRT wrapper (host/HavacHwClient.cpp:25-202) and the
// Vitis-HLS kernel it launches (device/HavacHls.cpp:20-43).  Host code here is
// thin: argument checks with the reference's own limits, hipMalloc/hipMemcpy,
// launches on a stream, HIP events.  There is no CPU fallback of any kind: if
// HIP or a gfx950 device is missing every entry point fails with
// HAVAC_E_NO_DEVICE / HAVAC_E_RUNTIME.
#include "../../include/havac_dev.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ssv_kernels.hip.h"
#include "hit_order.hip.h"

using namespace havac;

namespace {

std::string hip_msg(const char* what, hipError_t e) {
    return std::string(what) + ": " + hipGetErrorString(e);
}

#define HIP_TRY(ctx_err, expr)                                        \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) {                                       \
            (ctx_err) = hip_msg(#expr, _e);                           \
            return _e == hipErrorOutOfMemory ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME; \
        }                                                             \
    } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// ---- tiling of the diagonal axis ------------------------------------------
struct Tiling {
    uint32_t nrows_padded;
    int64_t first_diag;
    uint32_t ntiles;
};

Tiling make_tiling(uint64_t nsymbols, uint32_t nrows) {
    Tiling t;
    t.nrows_padded = round_up(nrows, kChunkRows);
    t.first_diag = -(int64_t)t.nrows_padded;   // <= -(nrows-1), multiple of 32
    uint64_t ndiag = nsymbols + t.nrows_padded;
    t.ntiles = (uint32_t)((ndiag + kTileDiags - 1) / kTileDiags);
    return t;
}

// Shards are runs of whole 12288-column segments: shard k of n owns columns [begin, end).  A shard computes
// the diagonals that reach its columns (a left halo of nrows-1 columns is recomputed) and
// reports only hits inside its columns, so the shards' ordered lists concatenate to the ordered whole.
void shard_columns(uint64_t nsymbols, uint32_t shard_index, uint32_t shard_count, uint64_t* begin, uint64_t* end) {
    const uint64_t nseg = nsymbols / HAVAC_SEGMENT_COLUMNS;
    *begin = nseg * shard_index / shard_count * HAVAC_SEGMENT_COLUMNS;
    *end = nseg * (shard_index + 1) / shard_count * HAVAC_SEGMENT_COLUMNS;
}

// tiles (runs of 2048 diagonals) that contain a cell of columns [begin, end)
void shard_tiles(const Tiling& t, uint32_t nrows, uint64_t col_begin, uint64_t col_end, uint32_t* tb, uint32_t* te) {
    if (col_end <= col_begin) { *tb = *te = 0; return; }
    // diagonal d = column - row, stored relative to first_diag = -nrows_padded
    const uint64_t lowest = col_begin + t.nrows_padded - (nrows - 1);      // column col_begin on the last row
    const uint64_t highest = col_end - 1 + t.nrows_padded;                 // column col_end-1 on row 0
    *tb = (uint32_t)(lowest / kTileDiags);
    *te = (uint32_t)(highest / kTileDiags) + 1;
    if (*te > t.ntiles) *te = t.ntiles;
}

// The columns the launch of shard [col_begin, col_end) READS: its own, the left halo of the diagonals that reach it
// (rows - 1 columns), and what the tiling adds at both ends -- a tile starts up to 2047 diagonals left of the first one
// needed, and a wave fetches the symbols of the chunk after the one it works on (ssv_diag_body: fetch_symbols).  Whole
// segments, clipped to the database.
void shard_window(uint64_t nsymbols, uint32_t nrows, uint64_t col_begin, uint64_t col_end, uint64_t* first, uint64_t* end) {
    const uint64_t reach = (uint64_t)round_up(nrows, kChunkRows) + 2 * kTileDiags;
    const uint64_t lo = col_begin > reach ? col_begin - reach : 0;
    const uint64_t hi = std::min<uint64_t>(nsymbols, col_end + 2 * kTileDiags);
    *first = lo / HAVAC_SEGMENT_COLUMNS * HAVAC_SEGMENT_COLUMNS;
    *end = std::min<uint64_t>(nsymbols, (hi + HAVAC_SEGMENT_COLUMNS - 1) / HAVAC_SEGMENT_COLUMNS * HAVAC_SEGMENT_COLUMNS);
}

// ---- partitions of a launch's units ----------------------------------------------------------------------------------------
struct PartitionKey {
    uint64_t nsymbols = 0; uint32_t nrows = 0; uint64_t col_begin = 0, col_end = 0; uint32_t tb = 0, te = 0, parts_log2 = 0;
    bool operator==(const PartitionKey& o) const {
        return nsymbols == o.nsymbols && nrows == o.nrows && col_begin == o.col_begin && col_end == o.col_end && tb == o.tb && te == o.te &&
               parts_log2 == o.parts_log2;
    }
};
// rows of tile `tile` (of the whole tiling) that the kernel walks: ssv_diag_body's p_lo, p_hi
inline uint64_t tile_rows(const Tiling& t, uint32_t tile, int64_t col_end) {
    const int64_t d0 = t.first_diag + (int64_t)tile * kTileDiags;
    const int64_t lo = std::min<int64_t>(std::max<int64_t>(-d0 - kTileDiags, 0), t.nrows_padded);
    const int64_t hi = std::min<int64_t>(std::max<int64_t>(col_end - d0, 0), t.nrows_padded);
    return hi > lo ? (uint64_t)(hi - lo) : 0u;
}
// part_begin[0 .. nparts]: tiles of the launch; partition k = tiles [part_begin[k], part_begin[k+1]) holds about 1/nparts of the rows
void plan_partitions(const Tiling& t, uint32_t tile_begin, uint32_t ntiles, int64_t col_end, uint32_t nparts, uint32_t* part_begin) {
    // (+ a constant per tile: its prologue and the step behind its last chunk -- what short tiles mostly consist of)
    auto tile_work = [&](uint32_t i) { return tile_rows(t, tile_begin + i, col_end) + 16; };
    uint64_t total = 0;
    for (uint32_t i = 0; i < ntiles; i++) total += tile_work(i);
    part_begin[0] = 0;
    uint64_t seen = 0;
    uint32_t k = 1;
    for (uint32_t i = 0; i < ntiles && k < nparts; i++) {
        seen += tile_work(i);
        while (k < nparts && seen * nparts >= total * k) part_begin[k++] = i + 1;
    }
    while (k <= nparts) part_begin[k++] = ntiles;
    part_begin[nparts] = ntiles;
}

// ---- row cuts of tall tiles ---------------------------------------------------------------------------------------------
// A cut tile is handed from wave to wave (ssv_kernels.hip.h, "work distribution"): every cut costs 2 x 4 KB through memory, and
// only the END of a launch needs fine grains.  So the blocks taper: each takes 1/guide of the rows that are left (whole
// chunk-flag words of 1024 rows), never less than `short_rows`, and a remainder below half of that joins the last block.
constexpr uint32_t kShortRows = 4096;       // finest row block
constexpr uint32_t kCutGuide = 2;           // a block takes 1/kCutGuide of the remaining rows
constexpr uint32_t kSplitRoundsX4 = 6;      // tiles of the last 1.5 rounds of wave slots are cut
constexpr uint32_t kChainRows = 16384;      // tallest uniform row block of a launch whose every tile is cut (fewer tiles than that many wave slots)
struct RowCuts { uint32_t ncuts = 0, uniform_rows = 0, nrow_blocks = 0; uint32_t cut[kMaxRowCuts + 1] = {}; };
RowCuts plan_row_cuts(uint32_t nrows_padded, uint32_t short_rows, uint32_t guide) {
    RowCuts r;
    short_rows = std::max(1024u, short_rows / 1024u * 1024u);
    guide = std::max(2u, guide);
    uint32_t at = 0;
    while (at < nrows_padded && r.ncuts < (uint32_t)kMaxRowCuts) {
        const uint32_t left = nrows_padded - at;
        uint32_t h = std::max(short_rows, (left / guide + 1023u) / 1024u * 1024u);
        if (left < h + short_rows / 2) h = left;
        at += h;
        r.cut[++r.ncuts] = std::min(at, nrows_padded);
    }
    // (what the table does not reach -- a slow taper over a tall model -- goes on in uniform blocks of `short_rows` rows)
    r.uniform_rows = short_rows;
    r.nrow_blocks = r.ncuts + (at < nrows_padded ? (nrows_padded - at + short_rows - 1) / short_rows : 0u);
    return r;
}

// ---- the plan of a launch -----------------------------------------------------------------------------------------------------
// Everything about HOW a launch hands out its tiles (ssv_kernels.hip.h, "items"), from the shape of the problem and the number
// of wave slots alone -- no device is touched: havac_ssv_plan (tests, tools) and havac_ssv_enqueue call the same function.
//  * eight partitions of adjacent tiles, one per XCD under the observed round-robin placement, of equal work;
//  * whole tiles first; where tiles are tall, the LAST tiles of every partition -- the last split_rounds rounds of wave slots
//    of the launch -- are cut by rows, the cuts getting finer towards the model's end (plan_row_cuts), so that the launch ends
//    everywhere at about the same time.  Round 2 cut EVERY tile into blocks of 8192 rows: the same balance, but a hand-off per
//    8192 x 2048 cells was 7-9 x the bytes the problem needs on C3 and C5.  (Cutting SHORT tiles to fill the last round -- C2
//    is 9.54 rounds -- does not pay: 2, 3, 4 row blocks per 1024-row tile took 2.20, 3.22, 4.5 ms against 2.02.);
//  * short models (at most kResidentRows rows: a tile is one to eight chunks and, handed out tile by tile, mostly prologue) take
//    ssv_resident_kernel: as many workgroups as the chip holds at once, the model's tables built once per workgroup and kept
//    in LDS, every wave walks its own run of adjacent tiles -- runs of equal length (+- one tile), dealt by the wave's number
//    alone (ssv_kernels.hip.h, "the resident-table variant").  No partitions, no tickets, no cut tiles, no block tails.
#ifndef HAVAC_WALK_SHARE_X16
#define HAVAC_WALK_SHARE_X16 12
#endif
constexpr uint64_t kWalkShareX16 = HAVAC_WALK_SHARE_X16;      // resident-table launches: a round's runs take 12/16 of the tiles that are left per wave slot
struct PlanTuning { int rows_per_block, tiles_per_item, parts_log2; uint32_t split_rounds_x4, short_rows, guide; int variant = -1; };
struct LaunchPlan { SsvRare L{}; uint32_t nblocks = 0, largest_item_rows = 0; bool resident_kernel = false; };
// which kernel runs a launch: the resident-table one where the model is short and nothing it leaves out is asked for (separator
// masks, the per-cell trace, a forced work distribution); `variant`: -1 = this rule, 0 = always the standard kernel, 1 = the
// resident-table one wherever it is valid (tests, A/B)
bool pick_resident_kernel(const Tiling& t, const PlanTuning& tune, bool has_mask, bool has_trace) {
    (void)has_mask;      // (round 5: served by ssv_resident_kernel_masked)
    if (tune.variant == 0 || has_trace || t.nrows_padded > kResidentRows) return false;
    return tune.variant == 1 || (tune.rows_per_block < 0 && tune.tiles_per_item == -1);
}
int plan_launch(std::string& err, const Tiling& t, uint32_t tb, uint32_t te, int64_t col_end, uint64_t slots, const PlanTuning& tune,
                const uint32_t* known_part_begin /* 9 entries of an earlier plan of the same shape, or null */, LaunchPlan& plan,
                bool resident_kernel = false) {
    SsvRare& L = plan.L;
    plan.resident_kernel = resident_kernel;
    L.first_diag = t.first_diag; L.tile_begin = tb; L.ntiles = te - tb;
    L.col_end = col_end;
    if (resident_kernel) {
        // Runs of adjacent tiles, handed out in ROUNDS of as many waves as the device holds (guided self-scheduling: the hardware
        // starts a round's workgroups as the slots of the round before come free).  A SIMD serves its oldest wave first, so the
        // waves of a round do not end together but one after the other; equal runs in ONE round left the SIMDs with five, four,
        // ... one wave over the last 60 % of the launch (50 TCUPS at 256 rows where single tiles run at 56).  So the runs taper:
        // a round takes 3/4 (kWalkShareX16: 9/16 and 14/16 measured within 1 %) of what is left per slot, the last rounds are single tiles -- a workgroup's start, its tables
        // and its end are paid ~3 times per slot instead of once per tile, and the launch still ends within a tile's time.
        // (tests: tiles_per_item = G >= 1 asks for runs of G tiles throughout, so that small problems walk too)
        const uint64_t nslots = std::max<uint64_t>(kWavesPerBlock, slots / kWavesPerBlock * kWavesPerBlock);
        L.parts_log2 = 0;
        for (uint32_t k = 0; k <= 8; k++) L.part_begin[k] = k ? L.ntiles : 0u;
        L.tiles_per_item = 1; L.single_tiles = 0; L.split_units = 0; L.nrow_blocks = 1;
        L.walk_slots = (uint32_t)nslots;
        uint64_t at = 0, nwaves = 0;
        uint32_t r = 0;
        for (; r < (uint32_t)kMaxWalkRounds && at < L.ntiles; r++) {
            const uint64_t left = L.ntiles - at;
            uint64_t len = tune.tiles_per_item >= 1 ? (uint64_t)tune.tiles_per_item : std::max<uint64_t>(1, left * kWalkShareX16 / (16 * nslots));
            if (r + 1 == (uint32_t)kMaxWalkRounds && tune.tiles_per_item < 1) len = std::max<uint64_t>(len, (left + nslots - 1) / nslots);   // (never reached in practice: the table's last round takes all that is left)
            L.walk_len[r] = (uint32_t)len; L.walk_base[r] = (uint32_t)at;
            const uint64_t want = (left + len - 1) / len;                      // waves that would finish the launch at this length
            const bool final_round = want <= nslots || len == 1 || tune.tiles_per_item >= 1 || r + 1 == (uint32_t)kMaxWalkRounds;
            const uint64_t waves = final_round ? (want + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock : nslots;
            nwaves += waves;
            at += waves * len;
            if (final_round) { r++; break; }
        }
        L.walk_rounds = std::max(1u, r);
        if (nwaves * 64 >= (1ull << 32)) { err = "too many tiles for one launch of runs of " + std::to_string(L.walk_len[0]) + " tiles"; return HAVAC_E_LENGTH; }
        plan.nblocks = L.ntiles ? (uint32_t)(nwaves / kWavesPerBlock) : 0u;
        plan.largest_item_rows = t.nrows_padded;
        return HAVAC_OK;
    }
    L.parts_log2 = L.ntiles >= 64 ? 3u : 0u;
    if (tune.parts_log2 >= 0) L.parts_log2 = (uint32_t)tune.parts_log2;
    const uint32_t nparts = 1u << L.parts_log2;
    RowCuts cuts{};
    uint32_t split_units = 0;
    if (te > tb && tune.rows_per_block < 0 && t.nrows_padded >= 2 * tune.short_rows) {
        split_units = (uint32_t)((tune.split_rounds_x4 * slots / 4 + nparts - 1) / nparts);
        if ((uint64_t)split_units * nparts >= L.ntiles) {
            // Fewer tiles than wave slots, or hardly more (C3 as stated: 5,130 tiles of 503,329 rows): every tile is a chain of row
            // blocks that runs from the launch's start to its end, and what keeps the chains in step is that the waves which WAIT
            // (there are more slots than chains) change places often.  Uniform blocks, then -- a taper that opens with half the rows
            // leaves the same SIMDs overbooked for half the launch (measured on C3: 97.3 ms against 89.6) -- of about 1/16 of the
            // model: blocks of 8192 rows (round 2) balance 0.5 % better than blocks of 16384 and move 1.5 x the bytes.
            const uint32_t rows = std::min(std::max((t.nrows_padded / 16 + 1023u) / 1024u * 1024u, 2 * tune.short_rows), kChainRows);
            cuts.ncuts = 0; cuts.cut[0] = 0; cuts.uniform_rows = rows;
            cuts.nrow_blocks = (t.nrows_padded + rows - 1) / rows;
            split_units = 0xffffffffu;
        } else {
            cuts = plan_row_cuts(t.nrows_padded, tune.short_rows, tune.guide);
        }
    } else if (te > tb && tune.rows_per_block > 0) {                       // experiments and tests (havac_ssv_set_tuning): uniform blocks, every tile cut
        const uint32_t rows_per_block = (uint32_t)tune.rows_per_block / 1024u * 1024u;    // whole chunk-flag words
        if (rows_per_block && rows_per_block < t.nrows_padded) {
            split_units = 0xffffffffu;
            cuts.ncuts = 0; cuts.cut[0] = 0; cuts.uniform_rows = rows_per_block;
            cuts.nrow_blocks = (t.nrows_padded + rows_per_block - 1) / rows_per_block;
        }
    }
    if (cuts.nrow_blocks < 2) split_units = 0;
    const bool split = split_units != 0;
    // Groups of adjacent tiles per wave (short models) are an experiment only: the waves of a round then walk in step, wait for
    // memory together and hide nothing of each other's prologues -- 4 tiles per wave 4-6 % slower than single tiles, 8 tiles
    // 10 %, with or without the next tile's symbols fetched a tile ahead.  tiles_per_item = G > 1: every tile in a
    // group; G < -1: groups of -G tiles, but a partition's last round of wave slots as single tiles.
    uint32_t tiles_per_item = 1, single_tiles = 0;
    const int walk = tune.tiles_per_item;
    if (walk >= 1 && !split) tiles_per_item = (uint32_t)walk;
    if (walk < -1 && !split) { tiles_per_item = (uint32_t)(-walk); single_tiles = (uint32_t)((slots + nparts - 1) / nparts); }
    L.tiles_per_item = tiles_per_item; L.single_tiles = single_tiles;
    plan.nblocks = 0;
    plan.largest_item_rows = t.nrows_padded;
    if (te <= tb) return HAVAC_OK;
    L.split_units = split_units;
    L.nrow_blocks = split ? cuts.nrow_blocks : 1u;
    L.ncuts = cuts.ncuts; L.uniform_rows = cuts.uniform_rows;
    for (uint32_t i = 0; i <= (uint32_t)kMaxRowCuts; i++) L.row_cut[i] = cuts.cut[i];
    // partitions: runs of tiles of about equal work (a tile's work = its rows inside the matrix and the shard's columns: the
    // tiles at the matrix's two ends are triangles -- with C3's 503,329 rows the first and the last eighth of the tiles would
    // hold 19 % less work than the others)
    if (known_part_begin) {
        for (uint32_t k = 0; k <= 8; k++) L.part_begin[k] = known_part_begin[k];
    } else {
        uint32_t begin[9];
        plan_partitions(t, tb, te - tb, col_end, nparts, begin);
        for (uint32_t k = 0; k <= 8; k++) L.part_begin[k] = begin[std::min(k, nparts)];
    }
    // blocks: every partition gets as many as its largest sibling needs (a block without items leaves at once)
    uint64_t most_items = 0;
    bool any_whole = false;
    for (uint32_t k = 0; k < nparts; k++) {
        const uint64_t mine = L.part_begin[k + 1] - L.part_begin[k], cut_units = std::min<uint64_t>(split_units, mine), whole = mine - cut_units;
        const uint64_t groups = tiles_per_item > 1 ? (whole - std::min<uint64_t>(single_tiles, whole)) / tiles_per_item : 0;
        most_items = std::max(most_items, groups + (whole - groups * tiles_per_item) + cut_units * L.nrow_blocks);
        any_whole = any_whole || whole > 0;
    }
    const uint64_t blocks64 = (most_items + kWavesPerBlock - 1) / kWavesPerBlock * nparts;
    if (blocks64 * (64 * kWavesPerBlock) >= (1ull << 32)) {      // a dispatch counts its work-items in 32 bits
        err = "too many tiles x row blocks for one launch (" + std::to_string(blocks64) + " workgroups)";
        return HAVAC_E_LENGTH;
    }
    plan.nblocks = (uint32_t)blocks64;
    if (split && !any_whole) {
        plan.largest_item_rows = cuts.uniform_rows;
        for (uint32_t i = 0; i < cuts.ncuts; i++) plan.largest_item_rows = std::max(plan.largest_item_rows, cuts.cut[i + 1] - cuts.cut[i]);
    }
    return HAVAC_OK;
}

// the control words of a context (havac_ssv_ctx::control)
constexpr uint32_t kControlTickets = 0, kControlFault = kTicketCounters * kTicketStride, kControlOrderState = (kTicketCounters + 1) * kTicketStride,
                   kControlWords = kControlOrderState + sizeof(OrderState) / sizeof(uint32_t);

// hand-off slots of a launch with cut tiles (ssv_kernels.hip.h, "items": slot = partition * split_units + unit, or the tile's
// number where every tile is cut)
inline size_t handoff_slots(const SsvRare& L) {
    if (L.split_units == 0) return 0;
    if (L.split_units == 0xffffffffu) return L.ntiles;
    return ((size_t)1 << L.parts_log2) * L.split_units;       // (< ntiles: plan_launch cuts every tile otherwise)
}

}  // namespace

// ===========================================================================
// Level 2: kernel ABI
// ===========================================================================
struct havac_ssv_ctx {
    int device = 0;
    uint32_t* rows8 = nullptr; size_t rows8_rows = 0;   // padded copy of the model
    uint32_t* chunk_flags = nullptr; size_t chunk_flag_words = 0;   // one byte per 32-row chunk, bit Q: window Q may test for hits at its end only
    // work distribution of the persistent kernel (ssv_kernels.hip.h, SsvLaunch)
    // the words a pass's kernels count in, cleared by ssv_prepare_model in front of every pass: [0, 144) the ticket counters of cut
    // tiles and the fault word, [144, 160) the ordering's OrderState (hit_order.hip.h)
    uint32_t* control = nullptr;
    uint32_t* block_flags = nullptr; size_t block_flag_tiles = 0;   // row-split launches: row blocks finished, per tile
    uint32_t* block_state = nullptr; size_t block_state_tiles = 0;  // row-split launches: 16 x 64 scores per tile
    uint64_t* tails = nullptr; uint32_t* tail_counts = nullptr; size_t tail_blocks = 0;   // block tails (ssv_kernels.hip.h): kTailSlots keys + a count per block
    int resident_blocks = 0;                                        // blocks the device holds at once (6 per CU)
    const uint16_t* pair_mask = nullptr;                // optional separator bitmap (boundary mode), caller-owned
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    uint64_t* sort_alt = nullptr; size_t sort_alt_count = 0;
    // the ordering (hit_order.hip.h, ssv_order_pass): per bucket a count / cursor and an offset, the list of large buckets
    uint32_t* bucket_counts = nullptr; uint32_t* bucket_offsets = nullptr; uint64_t* bucket_chunk_base = nullptr;
    uint32_t* bucket_large = nullptr; size_t bucket_alloc = 0;
    uint32_t large_capacity = kLargeBucketDefault;   // keys the ordering kernel's LDS sorter holds in the next launch (grown on demand)
    PassReport* h_report = nullptr;            // pinned: what the pass's ordering kernel tells the host
    OrderPass order_args{};                    // of the pending pass (a relaunch after growing a buffer reuses them)
    uint32_t order_tail_blocks = 0;            // blocks whose tails the pending pass's ordering gathers first (0: no tails in use)
    uint64_t last_found = 0;                   // records of the last finished pass: sizes the next pass's ordering grids
    bool tails_gathered = false;               // ... and whether that has been enqueued (a relaunch of the ordering must not gather twice)
    bool order_dirty = false;                  // an ordering was cut short: its counts cannot be trusted, the next pass starts from zeroed ones
    uint64_t first_segment = 0, nsegments = 0; // the shard's segments (bucket numbering)
    int tune_ordering = -1;                    // -1 / 1: bucket ordering, 0: always the radix sort (experiments, tests)
    uint32_t last_order_buckets = 0, last_order_largest = 0; int last_order_path = 0;   // what the last pass's ordering did (tests, tools)
    unsigned long long* d_count = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // pass start, kernel start (the model is prepared), kernel end, records ordered
    CellRecord* trace_cells = nullptr; uint32_t trace_row0 = 0, trace_rows = 0, trace_cols = 0; uint64_t trace_col0 = 0;   // per-cell trace window (debugging)
    hipStream_t order_stream = nullptr;        // optional: where finish() orders the records (default: the enqueue's stream)
    uint64_t window_first = 0, window_columns = 0;   // the caller's sequence buffer holds only these columns (0, 0: all of them)
    // experiment knobs (havac_ssv_set_tuning): -1 = the library decides
    int tune_rows_per_block = -1, tune_tiles_per_item = -1, tune_block_tails = -1;
    int tune_variant = -1;                     // havac_ssv_set_kernel_variant: -1 the library decides, 0 standard kernel, 1 resident-table kernel where valid
    bool last_resident_kernel = false;
    // havac_ssv_set_split_tuning: partitions (-1: the library decides), rounds of wave slots whose tiles are cut (x4), finest row block, taper
    PartitionKey part_key{}; uint32_t part_begin[9] = {};            // the partitions of the last launch's shape (plan_partitions)
    uint32_t last_plan_blocks = 0, last_plan_item_rows = 0;
    int tune_parts_log2 = -1; uint32_t tune_split_rounds_x4 = kSplitRoundsX4, tune_short_rows = kShortRows, tune_guide = kCutGuide;
    // the pass enqueue() started and finish() completes
    bool pending = false;                      // enqueue() done
    uint64_t found = 0;
    hipStream_t stream = nullptr;
    uint64_t* d_hits = nullptr;
    uint64_t hit_capacity = 0;
    unsigned key_bits = 64, row_bits = 24;             // significant bits / row-field width of the pending pass's sort keys
    float ssv_ms = 0.f, total_ms = 0.f;
    std::string err;
};

#ifdef HAVAC_WAVE_CLOCKS
extern "C" int havac_debug_wave_clocks(uint64_t* out, uint32_t nwaves) {      // experiments only: see g_wave_clocks
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_clocks), (size_t)nwaves * 4 * sizeof(uint64_t)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef HAVAC_SLOW_CLOCKS
extern "C" int havac_debug_slow_clocks(uint64_t* out, int reset) {      // experiments only: see g_slow_clocks
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slow_clocks), 4 * sizeof(uint64_t)) != hipSuccess) return 1;
    const uint64_t zero[4] = {0, 0, 0, 0};
    return reset && hipMemcpyToSymbol(HIP_SYMBOL(g_slow_clocks), zero, sizeof zero) != hipSuccess ? 1 : 0;
}
#endif
extern "C" const char* havac_dev_version(void) { return "havac_dev 0.1 gfx950"; }

extern "C" int havac_ssv_ctx_create(havac_ssv_ctx** out) {
    if (!out) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HAVAC_E_NO_DEVICE;
    havac_ssv_ctx* c = new (std::nothrow) havac_ssv_ctx;
    if (!c) return HAVAC_E_NOMEM;
    std::string err;
    auto fail = [&](int code) { havac_ssv_ctx_destroy(c); return code; };
    if (hipGetDevice(&c->device) != hipSuccess) return fail(HAVAC_E_NO_DEVICE);
    if (hipMalloc(&c->d_count, sizeof(unsigned long long)) != hipSuccess) return fail(HAVAC_E_NOMEM);
    if (hipMalloc(&c->control, kControlWords * sizeof(uint32_t)) != hipSuccess) return fail(HAVAC_E_NOMEM);
    if (hipMemset(c->control, 0, kControlWords * sizeof(uint32_t)) != hipSuccess) return fail(HAVAC_E_RUNTIME);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return fail(HAVAC_E_NO_DEVICE);
        c->resident_blocks = prop.multiProcessorCount * kBlocksPerCu;
    }
    if (hipHostMalloc(&c->h_report, sizeof(PassReport), hipHostMallocDefault) != hipSuccess) return fail(HAVAC_E_NOMEM);
    std::memset(c->h_report, 0, sizeof(PassReport));
    // (the ordering kernel's LDS sorter may be asked for 64 KB of dynamic LDS: above the default limit together with its few static words)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ssv_order_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLargeBucket * sizeof(uint32_t)));
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(HAVAC_E_RUNTIME);
    // the fills above ran on the null stream; passes run on streams of the caller's, possibly non-blocking ones
    if (hipDeviceSynchronize() != hipSuccess) return fail(HAVAC_E_RUNTIME);
    *out = c;
    return HAVAC_OK;
}

extern "C" void havac_ssv_ctx_destroy(havac_ssv_ctx* c) {
    if (!c) return;
    if (c->rows8) (void)hipFree(c->rows8);
    if (c->chunk_flags) (void)hipFree(c->chunk_flags);
    if (c->control) (void)hipFree(c->control);
    if (c->block_flags) (void)hipFree(c->block_flags);
    if (c->block_state) (void)hipFree(c->block_state);
    if (c->tails) (void)hipFree(c->tails);
    if (c->tail_counts) (void)hipFree(c->tail_counts);
    if (c->sort_tmp) (void)hipFree(c->sort_tmp);
    if (c->sort_alt) (void)hipFree(c->sort_alt);
    if (c->bucket_counts) (void)hipFree(c->bucket_counts);
    if (c->bucket_offsets) (void)hipFree(c->bucket_offsets);
    if (c->bucket_chunk_base) (void)hipFree(c->bucket_chunk_base);
    if (c->bucket_large) (void)hipFree(c->bucket_large);
    if (c->h_report) (void)hipHostFree(c->h_report);
    if (c->d_count) (void)hipFree(c->d_count);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

extern "C" const char* havac_ssv_ctx_last_error(havac_ssv_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" uint64_t havac_ssv_shard_cells(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index,
                                          uint32_t shard_count) {
    if (nrows == 0 || nsymbols == 0 || shard_count == 0 || shard_index >= shard_count) return 0;
    uint64_t b, e;
    shard_columns(nsymbols, shard_index, shard_count, &b, &e);
    return (e - b) * (uint64_t)nrows;
}

extern "C" int havac_ssv_set_separator_mask(havac_ssv_ctx* c, const uint8_t* d_pair_mask) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (((uintptr_t)d_pair_mask & 1u)) { c->err = "separator mask must be 2-byte aligned"; return HAVAC_E_ARGUMENT; }
    c->pair_mask = reinterpret_cast<const uint16_t*>(d_pair_mask);
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_cell_trace(havac_ssv_ctx* c, havac_cell_record* d_cells, uint32_t row0, uint64_t col0,
                                        uint32_t nrows, uint32_t ncols) {
    static_assert(sizeof(havac_cell_record) == sizeof(CellRecord) && sizeof(CellRecord) == 8, "one layout on both sides of the ABI");
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: set the trace window between passes"; return HAVAC_E_LOGIC; }
    if (d_cells && (nrows == 0 || ncols == 0)) { c->err = "empty trace window"; return HAVAC_E_ARGUMENT; }
    c->trace_cells = reinterpret_cast<CellRecord*>(d_cells);
    c->trace_row0 = row0; c->trace_col0 = col0; c->trace_rows = nrows; c->trace_cols = ncols;
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_order_stream(havac_ssv_ctx* c, void* hip_stream) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: set the ordering stream between passes"; return HAVAC_E_LOGIC; }
    c->order_stream = (hipStream_t)hip_stream;
    return HAVAC_OK;
}

extern "C" int havac_ssv_shard_window(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index, uint32_t shard_count,
                                      uint64_t* first_column, uint64_t* end_column) {
    if (!first_column || !end_column || shard_count == 0 || shard_index >= shard_count || nsymbols == 0 || nrows == 0 ||
        nsymbols % HAVAC_SEGMENT_COLUMNS != 0)
        return HAVAC_E_ARGUMENT;
    uint64_t b, e;
    shard_columns(nsymbols, shard_index, shard_count, &b, &e);
    shard_window(nsymbols, nrows, b, e, first_column, end_column);
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_sequence_window(havac_ssv_ctx* c, uint64_t first_column, uint64_t ncolumns) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: set the sequence window between passes"; return HAVAC_E_LOGIC; }
    if (first_column % HAVAC_SEGMENT_COLUMNS != 0 || ncolumns % HAVAC_SEGMENT_COLUMNS != 0) {
        c->err = "a sequence window is a run of whole 12288-column segments";
        return HAVAC_E_ARGUMENT;
    }
    c->window_first = first_column; c->window_columns = ncolumns;
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_tuning(havac_ssv_ctx* c, int rows_per_block, int tiles_per_item, int block_tails, int ordering) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: change the tuning between passes"; return HAVAC_E_LOGIC; }
    if (block_tails > 2 || ordering > 1 || (rows_per_block > 0 && rows_per_block < 1024)) { c->err = "bad tuning value"; return HAVAC_E_ARGUMENT; }
    c->tune_rows_per_block = rows_per_block; c->tune_tiles_per_item = tiles_per_item; c->tune_block_tails = block_tails;
    c->tune_ordering = ordering;
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_split_tuning(havac_ssv_ctx* c, int parts_log2, int split_rounds_x4, int short_rows, int guide) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: change the tuning between passes"; return HAVAC_E_LOGIC; }
    if (parts_log2 > 3 || (short_rows >= 0 && short_rows < 1024) || guide == 0 || guide == 1 || guide > 16) { c->err = "bad tuning value"; return HAVAC_E_ARGUMENT; }
    c->tune_parts_log2 = parts_log2 < 0 ? -1 : parts_log2;
    c->tune_split_rounds_x4 = split_rounds_x4 < 0 ? kSplitRoundsX4 : (uint32_t)split_rounds_x4;
    c->tune_short_rows = short_rows < 0 ? kShortRows : (uint32_t)short_rows / 1024u * 1024u;
    c->tune_guide = guide < 0 ? kCutGuide : (uint32_t)guide;
    return HAVAC_OK;
}

static int check_inputs(std::string& err, uint64_t nsymbols, uint32_t nrows);
static int ensure_alt(havac_ssv_ctx* c, uint64_t count, hipStream_t stream);
static int ensure_buckets(havac_ssv_ctx* c, uint32_t nbuckets, hipStream_t stream);
static int launch_ordering(havac_ssv_ctx* c, unsigned seg_bits);
extern "C" int havac_ssv_plan(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index, uint32_t shard_count, uint32_t wave_slots,
                              const int32_t* tuning, uint32_t ntuning, havac_launch_plan* out) {
    static_assert(kMaxRowCuts + 1 == 33, "havac_launch_plan::row_cut");
    if (!out || shard_count == 0 || shard_index >= shard_count || (!tuning && ntuning) || wave_slots > (1u << 20)) return HAVAC_E_ARGUMENT;
    std::string err;
    if (int rc = check_inputs(err, nsymbols, nrows)) return rc;
    int v[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    for (uint32_t i = 0; i < ntuning && i < 9; i++) v[i] = tuning[i];
    if ((v[0] > 0 && v[0] < 1024) || v[4] > 3 || (v[6] >= 0 && v[6] < 1024) || v[7] == 0 || v[7] == 1 || v[7] > 16 || v[8] > 1) return HAVAC_E_ARGUMENT;
    const PlanTuning tune{v[0], v[1], v[4] < 0 ? -1 : v[4], v[5] < 0 ? kSplitRoundsX4 : (uint32_t)v[5],
                          v[6] < 0 ? kShortRows : (uint32_t)v[6] / 1024u * 1024u, v[7] < 0 ? kCutGuide : (uint32_t)v[7], v[8] < 0 ? -1 : v[8]};
    const Tiling t = make_tiling(nsymbols, nrows);
    uint64_t col_begin, col_end;
    shard_columns(nsymbols, shard_index, shard_count, &col_begin, &col_end);
    uint32_t tb, te;
    shard_tiles(t, nrows, col_begin, col_end, &tb, &te);
    LaunchPlan plan;
    if (int rc = plan_launch(err, t, tb, te, (int64_t)col_end, wave_slots ? wave_slots : 256u * kBlocksPerCu * kWavesPerBlock, tune, nullptr, plan,
                             pick_resident_kernel(t, tune, false, false))) return rc;
    const SsvRare& L = plan.L;
    std::memset(out, 0, sizeof(*out));
    out->nrows_padded = t.nrows_padded; out->tile_begin = L.tile_begin; out->ntiles = L.ntiles;
    out->nparts = 1u << L.parts_log2;
    for (uint32_t k = 0; k <= 8; k++) out->part_begin[k] = L.part_begin[k];
    out->tiles_per_group = L.tiles_per_item; out->single_tiles = L.single_tiles; out->cut_tiles = L.split_units;
    out->nrow_blocks = L.ntiles ? L.nrow_blocks : 0; out->ncuts = L.ncuts; out->uniform_rows = L.uniform_rows;
    for (uint32_t i = 0; i <= (uint32_t)kMaxRowCuts; i++) out->row_cut[i] = L.row_cut[i];
    out->workgroups = plan.nblocks;
    out->resident_kernel = plan.resident_kernel ? 1u : 0u;
    out->walk_slots = L.walk_slots; out->walk_rounds = L.walk_rounds;
    for (int r = 0; r < kMaxWalkRounds; r++) { out->walk_len[r] = L.walk_len[r]; out->walk_base[r] = L.walk_base[r]; }
    return HAVAC_OK;
}

extern "C" int havac_ssv_set_kernel_variant(havac_ssv_ctx* c, int variant) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (c->pending) { c->err = "a pass is in flight: change the kernel variant between passes"; return HAVAC_E_LOGIC; }
    if (variant > 1) { c->err = "kernel variant: -1 (the library decides), 0 (standard) or 1 (resident tables: short models)"; return HAVAC_E_ARGUMENT; }
    c->tune_variant = variant < 0 ? -1 : variant;
    return HAVAC_OK;
}

extern "C" int havac_ssv_last_kernel_variant(havac_ssv_ctx* c, int* variant) {
    if (!c || !variant) return HAVAC_E_ARGUMENT;
    *variant = c->last_resident_kernel ? 1 : 0;
    return HAVAC_OK;
}

extern "C" int havac_ssv_wave_slots(havac_ssv_ctx* c, uint32_t* wave_slots) {
    if (!c || !wave_slots) return HAVAC_E_ARGUMENT;
    *wave_slots = (uint32_t)c->resident_blocks * kWavesPerBlock;
    return HAVAC_OK;
}

extern "C" int havac_ssv_last_ordering(havac_ssv_ctx* c, int* path, uint32_t* nbuckets, uint32_t* largest_bucket) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (path) *path = c->last_order_path;
    if (nbuckets) *nbuckets = c->last_order_buckets;
    if (largest_bucket) *largest_bucket = c->last_order_largest;
    return HAVAC_OK;
}

extern "C" int havac_ssv_shard_columns(uint64_t nsymbols, uint32_t shard_index, uint32_t shard_count,
                                       uint64_t* col_begin, uint64_t* col_end) {
    if (!col_begin || !col_end || shard_count == 0 || shard_index >= shard_count || nsymbols == 0 ||
        nsymbols % HAVAC_SEGMENT_COLUMNS != 0)
        return HAVAC_E_ARGUMENT;
    shard_columns(nsymbols, shard_index, shard_count, col_begin, col_end);
    return HAVAC_OK;
}

static int check_inputs(std::string& err, uint64_t nsymbols, uint32_t nrows) {
    // host/HavacHwClient.cpp:81-97,112-125,142-147; device/HavacHls.hpp:19-20
    if (nsymbols == 0) { err = "sequence length in segments cannot be 0, but 0 was given to the client."; return HAVAC_E_LENGTH; }
    if (nrows == 0) { err = "phmm length in vectors cannot be 0, but 0 was given to the client."; return HAVAC_E_LENGTH; }
    if (nsymbols % HAVAC_SEGMENT_COLUMNS != 0) {
        err = "sequence length must be a multiple of the sequence segment length 12288 but sequence of length " +
              std::to_string(nsymbols) + " was not";
        return HAVAC_E_LENGTH;
    }
    if (nsymbols / 4 >= (4ull << 30)) { err = "compressed sequence size must be less than 4GiB"; return HAVAC_E_LENGTH; }
    if (nrows >= (1u << 24)) { err = "model rows must be below 2^24 (24-bit row field of the hit record)"; return HAVAC_E_LENGTH; }
    return HAVAC_OK;
}

extern "C" int havac_ssv_enqueue(havac_ssv_ctx* c, const uint8_t* d_sequence, uint64_t nsymbols,
                                 const int8_t* d_phmm, uint32_t nrows, uint32_t shard_index,
                                 uint32_t shard_count, uint64_t* d_hits, uint64_t hit_capacity,
                                 const uint32_t* d_abort_flag, void* hip_stream) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (!d_sequence || !d_phmm || (!d_hits && hit_capacity) || shard_count == 0 || shard_index >= shard_count) {
        c->err = "bad argument to havac_ssv_enqueue";
        return HAVAC_E_ARGUMENT;
    }
    if (((uintptr_t)d_sequence & 7u) || ((uintptr_t)d_phmm & 3u)) {
        c->err = "sequence must be 8-byte aligned and model 4-byte aligned";
        return HAVAC_E_ARGUMENT;
    }
    if (c->pending) { c->err = "previous pass not finished: call havac_ssv_finish first"; return HAVAC_E_LOGIC; }
    int rc = check_inputs(c->err, nsymbols, nrows);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    HIP_TRY(c->err, hipSetDevice(c->device));

    Tiling t = make_tiling(nsymbols, nrows);
    const uint32_t model_words = t.nrows_padded + kModelSlack;   // "row -1", the rows, padding rows
    if (c->rows8_rows < model_words) {
        if (c->rows8) (void)hipFree(c->rows8);
        c->rows8 = nullptr; c->rows8_rows = 0;
        HIP_TRY(c->err, hipMalloc(&c->rows8, (size_t)model_words * sizeof(uint32_t)));
        c->rows8_rows = model_words;
    }
    const uint32_t flag_words = (t.nrows_padded / kChunkRows + 1 + 3) / 4 + 2;     // a byte per chunk; + 2: the kernels load a word ahead
    if (c->chunk_flag_words < flag_words) {
        if (c->chunk_flags) (void)hipFree(c->chunk_flags);
        c->chunk_flags = nullptr; c->chunk_flag_words = 0;
        HIP_TRY(c->err, hipMalloc(&c->chunk_flags, (size_t)flag_words * sizeof(uint32_t)));
        c->chunk_flag_words = flag_words;
    }
    uint64_t col_begin, col_end;
    shard_columns(nsymbols, shard_index, shard_count, &col_begin, &col_end);
    uint32_t tb, te;
    shard_tiles(t, nrows, col_begin, col_end, &tb, &te);
    if (c->window_columns) {
        // the caller's buffer holds columns [window_first, window_first + window_columns) only: it must cover what this
        // shard's launch reads; the kernel is handed the address column 0 would have
        uint64_t need_first, need_end;
        shard_window(nsymbols, nrows, col_begin, col_end, &need_first, &need_end);
        if (c->window_first > need_first || c->window_first + c->window_columns < need_end) {
            c->err = "the sequence window [" + std::to_string(c->window_first) + ", " + std::to_string(c->window_first + c->window_columns) +
                     ") does not cover the columns this shard reads [" + std::to_string(need_first) + ", " + std::to_string(need_end) + ")";
            return HAVAC_E_ARGUMENT;
        }
        d_sequence -= c->window_first / 4;
    }
    const uint16_t* const pair_mask = c->pair_mask ? c->pair_mask - (c->window_columns ? c->window_first / 32 : 0) : nullptr;

    // sort key = segment | row | column in segment, each field only as wide as this problem needs
    unsigned row_bits = 1, seg_bits = 1;
    while ((1u << row_bits) < t.nrows_padded + 2u) row_bits++;
    while ((1ull << seg_bits) < nsymbols / HAVAC_SEGMENT_COLUMNS) seg_bits++;
    // ---- how the tiles are handed out (plan_launch; ssv_kernels.hip.h, "items") ----
    SsvRare L{};          // the kernel's first argument: tiling, hit queue, hand-off buffers (read from the kernarg segment on demand)
    const uint64_t slots = (uint64_t)c->resident_blocks * kWavesPerBlock;
    const PlanTuning tuning{c->tune_rows_per_block, c->tune_tiles_per_item, c->tune_parts_log2, c->tune_split_rounds_x4, c->tune_short_rows, c->tune_guide, c->tune_variant};
    const bool resident_kernel = pick_resident_kernel(t, tuning, c->pair_mask != nullptr, c->trace_cells != nullptr);
    {
        // (the partitions are kept from pass to pass while the shape stays the same: a pass of C2 is 1.9 ms, the table takes 50,000 tiles)
        const PartitionKey key{nsymbols, nrows, col_begin, col_end, tb, te, (uint32_t)c->tune_parts_log2};
        const bool cached = key == c->part_key;
        LaunchPlan plan;
        if (int prc = plan_launch(c->err, t, tb, te, (int64_t)col_end, slots, tuning, cached && !resident_kernel ? c->part_begin : nullptr, plan, resident_kernel)) return prc;
        if (!cached && !resident_kernel) { for (uint32_t k = 0; k <= 8; k++) c->part_begin[k] = plan.L.part_begin[k]; c->part_key = key; }
        L = plan.L;
        c->last_plan_blocks = plan.nblocks; c->last_plan_item_rows = plan.largest_item_rows;
    }
    const uint32_t nblocks = c->last_plan_blocks, largest_item_rows = c->last_plan_item_rows;
    const bool split = L.split_units != 0;
    // ---- the buffers a pass needs, BEFORE anything is enqueued: growing one waits for what the stream still holds -------------------
    size_t handoff_words = 0;
    if (te > tb && split) {
        // one hand-off slot (a count word + 2 KB of scores) per CUT tile -- the last split_units tiles of every partition, or
        // every tile -- not per tile of the launch (a 3 Gbp genome against a tall model is 1.5 M tiles of
        // which ~9,000 are cut; the counts are cleared on every pass, by ssv_prepare_model)
        const size_t slots_needed = handoff_slots(L);
        if (c->block_flag_tiles < slots_needed) {
            HIP_TRY(c->err, hipStreamSynchronize(stream));        // an earlier pass on this stream may still use the old buffers
            if (c->block_flags) (void)hipFree(c->block_flags);
            if (c->block_state) (void)hipFree(c->block_state);
            c->block_flags = c->block_state = nullptr; c->block_flag_tiles = c->block_state_tiles = 0;
            HIP_TRY(c->err, hipMalloc(&c->block_flags, slots_needed * sizeof(uint32_t)));
            HIP_TRY(c->err, hipMalloc(&c->block_state, slots_needed * (kRegs / 2) * 64 * sizeof(uint32_t)));
            c->block_flag_tiles = c->block_state_tiles = slots_needed;
        }
        handoff_words = slots_needed;
    }
    // block tails: a side buffer of kTailSlots keys per block, where blocks are short-lived -- items of up to 512 rows: a
    // block of C2 (1024 rows) lives 200 us and loses nothing to the one atomic at its end, while its tail would cross
    // HBM three times instead of once (measured: the same step time, 43.8 against 39.2 MB of traffic per launch) -- and
    // the buffer stays below 128 MB
    bool use_tails = te > tb && largest_item_rows <= 512 && (uint64_t)nblocks * kTailSlots * sizeof(uint64_t) <= (128ull << 20);
    if (te > tb && c->tune_block_tails >= 0)      // experiments: 0 = off, 1 = the default rule, 2 = on whatever the height of an item
        use_tails = c->tune_block_tails == 2 ? (uint64_t)nblocks * kTailSlots * sizeof(uint64_t) <= (128ull << 20) : use_tails && c->tune_block_tails != 0;
    if (use_tails && c->tail_blocks < nblocks) {
        HIP_TRY(c->err, hipStreamSynchronize(stream));        // an earlier pass on this stream may still use the old buffers
        if (c->tails) (void)hipFree(c->tails);
        if (c->tail_counts) (void)hipFree(c->tail_counts);
        c->tails = nullptr; c->tail_counts = nullptr; c->tail_blocks = 0;
        const size_t want = (size_t)nblocks + nblocks / 4 + 64;
        HIP_TRY(c->err, hipMalloc(&c->tails, want * kTailSlots * sizeof(uint64_t)));
        HIP_TRY(c->err, hipMalloc(&c->tail_counts, want * sizeof(uint32_t)));
        c->tail_blocks = want;
    }
    // the ordering's second buffer and bucket tables: a first guess here, grown by finish() when the kernel says a pass needs more
    if (hit_capacity) {
        if (int rc2 = ensure_alt(c, std::min<uint64_t>(hit_capacity, 1ull << 22), stream)) return rc2;
        if (int rc2 = ensure_buckets(c, 1u << 16, stream)) return rc2;
    }
    if (c->order_dirty && c->bucket_counts) {     // an earlier ordering was cut short: its counts cannot be trusted
        HIP_TRY(c->err, hipMemsetAsync(c->bucket_counts, 0, c->bucket_alloc * sizeof(uint32_t), stream));
        c->order_dirty = false;
    }

    HIP_TRY(c->err, hipEventRecord(c->ev[0], stream));
    // one launch: the padded copy of the model, the chunk flags, and everything the pass's kernels count in, cleared -- the hit
    // counter, the tickets, the tails' sums, the ordering's barrier words, the hand-off counts of cut tiles.  (Separator pairs
    // score -128 twice in a row whatever the model says: with a mask the kernel takes a chunk's flag only where no separator and no
    // edge of the matrix lies in the wave's window, see ssv_diag_body.)
    {
        const uint32_t nflagwords = flag_words;
        const uint32_t threads = std::max(std::max(model_words, nflagwords * 4u), kControlWords);
        hipLaunchKernelGGL(ssv_prepare_model, dim3((threads + 255) / 256), dim3(256), 0, stream, d_phmm, nrows, c->rows8, model_words,
                           t.nrows_padded, c->chunk_flags, nflagwords, c->d_count, c->control, kControlWords, c->block_flags, (uint32_t)handoff_words);
    }
    HIP_TRY(c->err, hipEventRecord(c->ev[1], stream));
    if (te > tb) {
        SsvRare& R = L;
        R.hits = d_hits; R.hit_count = c->d_count; R.hit_capacity = hit_capacity;
        R.col_begin = (int64_t)col_begin; R.col_span = col_end - col_begin;
        R.half_tail = (resident_kernel && t.nrows_padded - nrows >= (uint32_t)kChunkRows / 2) ? 1u : 0u;
        R.abort_flag = d_abort_flag; R.pair_mask = pair_mask; R.tickets = c->control + kControlTickets; R.block_flags = c->block_flags; R.block_state = c->block_state;
        R.fault = c->control + kControlFault; R.row_bits = row_bits;
        R.tails = use_tails ? c->tails : nullptr; R.tail_counts = use_tails ? c->tail_counts : nullptr;
        R.cells = c->trace_cells; R.cell_row0 = c->trace_row0; R.cell_col0 = (int64_t)c->trace_col0;
        R.cell_rows = c->trace_rows; R.cell_cols = c->trace_cols;
        // (round 5: the chunk flags are handed over with a separator mask too -- a chunk then tests every two steps only where a
        // separator, or the matrix's edge, really lies in the wave's window; the kernel sees the mask in SsvRare::pair_mask)
        const uint32_t* const safe_chunks = (const uint32_t*)c->chunk_flags;
        c->last_resident_kernel = resident_kernel;
        if (c->trace_cells)       // debugging: the same kernel body with the per-cell trace compiled in
            hipLaunchKernelGGL(ssv_diag_kernel_traced, dim3(nblocks), dim3(64 * kWavesPerBlock), 0, stream, R,
                               d_sequence, (const uint32_t*)c->rows8, safe_chunks, (int64_t)nsymbols, t.nrows_padded);
        else if (resident_kernel && pair_mask)
            hipLaunchKernelGGL(ssv_resident_kernel_masked, dim3(nblocks), dim3(64 * kWavesPerBlock), 0, stream, R,
                               d_sequence, (const uint32_t*)c->rows8, safe_chunks, (int64_t)nsymbols, t.nrows_padded);
        else if (resident_kernel)    // short models: the chip's worth of workgroups, tables resident in LDS, every wave walks its run of tiles
            hipLaunchKernelGGL(ssv_resident_kernel, dim3(nblocks), dim3(64 * kWavesPerBlock), 0, stream, R,
                               d_sequence, (const uint32_t*)c->rows8, safe_chunks, (int64_t)nsymbols, t.nrows_padded);
        else
            hipLaunchKernelGGL(ssv_diag_kernel, dim3(nblocks), dim3(64 * kWavesPerBlock), 0, stream, R,
                               d_sequence, (const uint32_t*)c->rows8, safe_chunks, (int64_t)nsymbols, t.nrows_padded);
    }
    HIP_TRY(c->err, hipEventRecord(c->ev[2], stream));
    // The third and last launch of the pass, the ordering (hit_order.hip.h: the count is read on the device, the tails where they
    // lie), runs on the ordering stream when there is one: the kernel stream is then free for the next pass's kernel at once.
    c->pending = true; c->stream = stream; c->d_hits = d_hits; c->hit_capacity = hit_capacity;
    c->row_bits = row_bits;
    c->key_bits = 14 + row_bits + seg_bits;
    c->first_segment = col_begin / HAVAC_SEGMENT_COLUMNS; c->nsegments = (col_end - col_begin) / HAVAC_SEGMENT_COLUMNS;
    c->order_tail_blocks = use_tails ? nblocks : 0u; c->tails_gathered = false;
    const hipStream_t after = c->order_stream ? c->order_stream : stream;
    if (after != stream) { if (hipError_t e = hipStreamWaitEvent(after, c->ev[2], 0); e != hipSuccess) { c->pending = false; c->err = hip_msg("hipStreamWaitEvent", e); return HAVAC_E_RUNTIME; } }
    if (int orc = launch_ordering(c, seg_bits)) { c->pending = false; return orc; }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { c->pending = false; c->err = hip_msg("launching the pass", e); return HAVAC_E_RUNTIME; }
    return HAVAC_OK;
}

// rocPRIM's default switches from merge sort to Onesweep radix sort above 1 Mi keys; C2's ~1.0 M hits sit just
// below and took 21 launches.  Onesweep from 64 Ki keys on, and sort keys only as wide as the problem needs (C2: 37 bits = 5 passes).  Measured on C2: 0.19 ms -> 0.166 ms from kernel end to ordered records.
using hit_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 64 * 1024>;

// radix sort of `count` keys held in `keys` (in place via the alt buffer), on `stream`; only the low `key_bits`
// bits can differ between keys
static int sort_run(havac_ssv_ctx* c, uint64_t* keys, uint64_t* alt, uint64_t count, hipStream_t stream, unsigned key_bits) {
    // `count` keys in `keys` -> sorted in `alt`
    size_t need = 0;
    HIP_TRY(c->err, rocprim::radix_sort_keys<hit_sort_config>(nullptr, need, keys, alt, (size_t)count, 0, key_bits, stream));
    if (c->sort_tmp_bytes < need) {
        HIP_TRY(c->err, hipStreamSynchronize(stream));        // an earlier run on this stream may still use the old one
        if (c->sort_tmp) (void)hipFree(c->sort_tmp);
        c->sort_tmp = nullptr; c->sort_tmp_bytes = 0;
        HIP_TRY(c->err, hipMalloc(&c->sort_tmp, need + need / 4));
        c->sort_tmp_bytes = need + need / 4;
    }
    size_t bytes = c->sort_tmp_bytes;
    HIP_TRY(c->err, rocprim::radix_sort_keys<hit_sort_config>(c->sort_tmp, bytes, keys, alt, (size_t)count, 0, key_bits, stream));
    return HAVAC_OK;
}

static int sort_keys(havac_ssv_ctx* c, uint64_t* keys, uint64_t count, hipStream_t stream, unsigned key_bits = 64) {
    if (count < 2) return HAVAC_OK;
    if (int rc = ensure_alt(c, count, stream)) return rc;
    // (rocPRIM orders more than 2^32 keys in one call: 4.5e9 checked, tools/big_sort_check.py)
    int rc = sort_run(c, keys, c->sort_alt, count, stream, key_bits);
    if (rc) return rc;
    HIP_TRY(c->err, hipMemcpyAsync(keys, c->sort_alt, count * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    return HAVAC_OK;
}

// ---- ordering ------------------------------------------------------------------------------------------------------------
static int ensure_alt(havac_ssv_ctx* c, uint64_t count, hipStream_t stream) {
    if (c->sort_alt_count >= count) return HAVAC_OK;
    HIP_TRY(c->err, hipStreamSynchronize(stream));            // an earlier pass on this stream may still use the old one
    if (c->sort_alt) (void)hipFree(c->sort_alt);
    c->sort_alt = nullptr; c->sort_alt_count = 0;
    const size_t want = (size_t)count + count / 4 + 1024;
    HIP_TRY(c->err, hipMalloc(&c->sort_alt, want * sizeof(uint64_t)));
    c->sort_alt_count = want;
    return HAVAC_OK;
}

static int ensure_buckets(havac_ssv_ctx* c, uint32_t nbuckets, hipStream_t stream) {
    if (c->bucket_alloc >= nbuckets) return HAVAC_OK;
    HIP_TRY(c->err, hipStreamSynchronize(stream));
    if (c->bucket_counts) (void)hipFree(c->bucket_counts);
    if (c->bucket_offsets) (void)hipFree(c->bucket_offsets);
    if (c->bucket_chunk_base) (void)hipFree(c->bucket_chunk_base);
    if (c->bucket_large) (void)hipFree(c->bucket_large);
    c->bucket_counts = nullptr; c->bucket_offsets = nullptr; c->bucket_chunk_base = nullptr; c->bucket_large = nullptr; c->bucket_alloc = 0;
    const size_t want = (size_t)nbuckets + nbuckets / 4 + 1024;
    HIP_TRY(c->err, hipMalloc(&c->bucket_counts, want * sizeof(uint32_t)));
    HIP_TRY(c->err, hipMalloc(&c->bucket_offsets, want * sizeof(uint32_t)));
    HIP_TRY(c->err, hipMalloc(&c->bucket_chunk_base, (want / kScanChunk + 2) * sizeof(uint64_t)));
    HIP_TRY(c->err, hipMalloc(&c->bucket_large, want * sizeof(uint32_t)));
    // once: every pass leaves the counts at zero.  (On the pass's own stream: a null-stream hipMemset is not ordered against a
    // non-blocking stream, and the kernels would race with it.)
    HIP_TRY(c->err, hipMemsetAsync(c->bucket_counts, 0, want * sizeof(uint32_t), stream));
    c->bucket_alloc = want;
    return HAVAC_OK;
}

// the generic path: radix sort of the keys, then key -> record
static int order_by_radix_sort(havac_ssv_ctx* c, uint64_t* d_hits, uint64_t count, hipStream_t stream) {
    int rc = sort_keys(c, d_hits, count, stream, c->key_bits);
    if (rc) return rc;
    if (count)
        hipLaunchKernelGGL(ssv_keys_to_records, dim3((unsigned)std::min<uint64_t>((count + 255) / 256, 1u << 20)), dim3(256), 0,
                           stream, d_hits, count, c->row_bits);
    return HAVAC_OK;
}

// The pass's ordering kernel (hit_order.hip.h: ssv_order_pass), enqueued behind the SSV kernel with what the context holds NOW;
// finish() launches it again when the kernel reports that a buffer was too small for this pass.  A persistent grid that is
// resident at once: one workgroup per CU where passes run beside each other (what an SSV kernel leaves free), more when the
// pass has the chip to itself.
static int launch_ordering(havac_ssv_ctx* c, unsigned seg_bits) {
    const hipStream_t after = c->order_stream ? c->order_stream : c->stream;
    OrderPass& a = c->order_args;
    a.hits = c->d_hits; a.alt = c->sort_alt; a.alt_capacity = c->sort_alt_count; a.capacity = c->hit_capacity;
    a.hit_count = c->d_count;
    a.counts = c->bucket_counts; a.offsets = c->bucket_offsets; a.chunk_base = c->bucket_chunk_base; a.large_list = c->bucket_large;
    a.max_buckets = (uint32_t)std::min<size_t>(c->bucket_alloc, 1u << 27);
    a.state = reinterpret_cast<OrderState*>(c->control + kControlOrderState);
    a.host = c->h_report; a.ssv_fault = c->control + kControlFault;
    a.row_bits = c->row_bits; a.seg_bits = seg_bits; a.large_capacity = c->large_capacity;
    a.generic = (c->tune_ordering == 0 || c->hit_capacity == 0 || !c->sort_alt || !c->bucket_counts) ? 1u : 0u;
    a.first_segment = c->first_segment; a.nsegments = c->nsegments;
    // what the blocks left in their tails joins the queue first (short items only; ssv_kernels.hip.h, "block tails")
    if (c->order_tail_blocks && !c->tails_gathered) {
        hipLaunchKernelGGL(ssv_gather_tails, dim3((c->order_tail_blocks + kGatherBlocks - 1) / kGatherBlocks), dim3(256), 0, after, (const uint64_t*)c->tails,
                           (const uint32_t*)c->tail_counts, c->order_tail_blocks, c->d_hits, c->d_count, c->hit_capacity);
        c->tails_gathered = true;
    }
    const size_t scan_bytes = (size_t)(kScanChunk + 64u) * sizeof(uint32_t), large_bytes = (size_t)c->large_capacity * sizeof(uint32_t);
    const int cus = c->resident_blocks / kBlocksPerCu;
    c->h_report->status = kOrderNone;
    // Grids: the kernels walk the list in strides of their grid, so any size is right; the count is not known here, the LAST
    // pass's is (same context, usually the same workload): enough workgroups for a quarter more than that, within bounds --
    // a grid whose workgroups mostly have nothing to do still costs its dispatch (a 64-row model x 100 Mbp orders 50,000 records).
    const uint64_t guess = c->last_found ? c->last_found + c->last_found / 4 + 4096 : ~0ull;
    const dim3 block(kOrderThreads);
    const dim3 stream_grid((unsigned)std::min<uint64_t>(12 * cus, std::max<uint64_t>(16, guess / 1024)));      // 1024 keys per workgroup and step
    const dim3 sort_grid((unsigned)std::min<uint64_t>(32 * cus, std::max<uint64_t>(16, guess / (kTargetBucket / 2 * 4))));     // a bucket per wave: every sort is a chain of memory latencies
    hipLaunchKernelGGL(ssv_order_count, stream_grid, block, 0, after, a);
    hipLaunchKernelGGL(ssv_order_scan, dim3((unsigned)((a.max_buckets + kScanChunk - 1) / kScanChunk)), block, scan_bytes, after, a);
    hipLaunchKernelGGL(ssv_order_scatter, stream_grid, block, 0, after, a);
    // (the large buckets' LDS sorter has a launch of its own: the three kernels that touch every record then fit into what an SSV
    // kernel leaves free on a SIMD, and only its few workgroups take the ticket that finds the one which reports)
    hipLaunchKernelGGL(ssv_order_sort, sort_grid, block, 0, after, a);
    hipLaunchKernelGGL(ssv_order_finish, dim3((unsigned)std::max(1, cus / 2)), block, large_bytes, after, a);
    HIP_TRY(c->err, hipGetLastError());
    HIP_TRY(c->err, hipEventRecord(c->ev[3], after));
    return HAVAC_OK;
}

// finish = begin + end.  Since round 5 a pass is enqueued whole -- preparation, SSV kernel, ordering: three launches, the count
// never visits the host in between -- so _begin has nothing left to do; the two halves stay for callers that begin several
// contexts before they end any (havac_dev_wait over the GPUs of a handle).
extern "C" int havac_ssv_finish_begin(havac_ssv_ctx* c) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (!c->pending) { c->err = "no pass enqueued"; return HAVAC_E_LOGIC; }
    return HAVAC_OK;
}

extern "C" int havac_ssv_finish_end(havac_ssv_ctx* c, uint64_t* hit_count_out) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (!c->pending) { c->err = "no pass enqueued"; return HAVAC_E_LOGIC; }
    // (every return below ends the pass)
    c->pending = false;
    HIP_TRY(c->err, hipSetDevice(c->device));
    const hipStream_t after = c->order_stream ? c->order_stream : c->stream;
    const PassReport* const rep = c->h_report;
    c->last_order_path = 1; c->last_order_buckets = 0; c->last_order_largest = 0;
    for (int attempt = 0;; attempt++) {
        if (hipError_t e = hipEventSynchronize(c->ev[3]); e != hipSuccess) { c->order_dirty = true; c->err = hip_msg("hipEventSynchronize", e); return HAVAC_E_RUNTIME; }
        const uint32_t status = __atomic_load_n(&rep->status, __ATOMIC_ACQUIRE);
        c->found = rep->found;
        c->last_order_buckets = rep->nbuckets; c->last_order_largest = rep->largest;
        if (rep->ssv_fault) {
            c->order_dirty = true;
            c->err = "the SSV kernel gave up waiting for a row block of a tile (work-queue fault)";
            return HAVAC_E_RUNTIME;
        }
        if (status == kOrderDone || status == kOrderOverflow) { c->last_found = std::min<uint64_t>(c->found, c->hit_capacity); break; }
        if (status == kOrderRetry && attempt < 4) {
            // a buffer of the context was too small for this pass; nothing was moved: grow it, clear the barrier words, again
            if (int rc = ensure_alt(c, rep->need_alt, after)) return rc;
            if (rep->need_buckets == 0 || rep->need_buckets > c->bucket_alloc) { if (int rc = ensure_buckets(c, std::max<uint32_t>(rep->need_buckets, 1u << 16), after)) return rc; }
            if (rep->need_large > c->large_capacity) { uint32_t cap = c->large_capacity; while (cap < rep->need_large && cap < kLargeBucket) cap *= 2; c->large_capacity = cap; }
            HIP_TRY(c->err, hipMemsetAsync(c->control + kControlOrderState, 0, sizeof(OrderState), after));
            unsigned seg_bits = c->key_bits - 14 - c->row_bits;
            if (int rc = launch_ordering(c, seg_bits)) return rc;
            continue;
        }
        if (status == kOrderGeneric) {
            // the generic path: what the blocks left in their tails joins the queue, then radix sort + key -> record
            c->last_order_path = c->tune_ordering == 0 ? 0 : 2;
            if (rep->oversized & 2u) HIP_TRY(c->err, hipMemsetAsync(c->bucket_counts, 0, c->bucket_alloc * sizeof(uint32_t), after));
            if (int rc = order_by_radix_sort(c, c->d_hits, c->found, after)) return rc;
            HIP_TRY(c->err, hipEventRecord(c->ev[3], after));
            HIP_TRY(c->err, hipEventSynchronize(c->ev[3]));
            break;
        }
        c->order_dirty = true;
        c->err = "the ordering kernels did not report (status " + std::to_string(status) + ")";
        return HAVAC_E_RUNTIME;
    }
    HIP_TRY(c->err, hipEventElapsedTime(&c->ssv_ms, c->ev[1], c->ev[2]));
    HIP_TRY(c->err, hipEventElapsedTime(&c->total_ms, c->ev[0], c->ev[3]));
    if (hit_count_out) *hit_count_out = c->found;
    if (c->found > c->hit_capacity) {
        c->err = "hit buffer overflow: " + std::to_string(c->found) + " hits found, capacity " + std::to_string(c->hit_capacity);
        return HAVAC_E_HIT_OVERFLOW;
    }
    return HAVAC_OK;
}

extern "C" int havac_ssv_finish(havac_ssv_ctx* c, uint64_t* hit_count_out) {
    const int rc = havac_ssv_finish_begin(c);
    if (rc) return rc;
    return havac_ssv_finish_end(c, hit_count_out);
}

extern "C" int havac_ssv_query(havac_ssv_ctx* c) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (!c->pending) { c->err = "no pass enqueued"; return HAVAC_E_LOGIC; }
    const hipError_t q = hipEventQuery(c->ev[3]);
    if (q == hipSuccess) return 1;
    if (q == hipErrorNotReady) return 0;
    c->err = hip_msg("hipEventQuery", q);
    return HAVAC_E_RUNTIME;
}

extern "C" int havac_ssv_wait_inputs(havac_ssv_ctx* c, int sequence_too) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (!c->pending) return HAVAC_OK;                      // nothing enqueued reads anything
    HIP_TRY(c->err, hipSetDevice(c->device));
    HIP_TRY(c->err, hipEventSynchronize(c->ev[sequence_too ? 2 : 1]));
    return HAVAC_OK;
}

extern "C" int havac_ssv_sort_hits(havac_ssv_ctx* c, uint64_t* d_hits, uint64_t count, void* hip_stream) {
    if (!c || (!d_hits && count)) return HAVAC_E_ARGUMENT;
    if (count == 0) return HAVAC_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    HIP_TRY(c->err, hipSetDevice(c->device));
    unsigned nb = (unsigned)std::min<uint64_t>((count + 255) / 256, 1u << 20);     // grid-stride kernels
    hipLaunchKernelGGL(ssv_records_to_keys, dim3(nb), dim3(256), 0, stream, d_hits, count, 24u);
    int rc = sort_keys(c, d_hits, count, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ssv_keys_to_records, dim3(nb), dim3(256), 0, stream, d_hits, count, 24u);
    HIP_TRY(c->err, hipGetLastError());
    return HAVAC_OK;
}

// The check of a gathered list (hit_order.hip.h: ssv_check_order): device order without duplicates, every rank's records
// inside that rank's columns.  Works on the device that is current; one pass, 40 bytes + 3 small tables of its own.
extern "C" int havac_ssv_check_order(const uint64_t* d_records, uint64_t count, const uint64_t* rank_counts,
                                     const uint64_t* span_begin, const uint64_t* span_end, uint32_t nranks, void* hip_stream,
                                     havac_order_report* out) {
    static_assert(sizeof(havac_order_report) == sizeof(OrderReport), "one layout on both sides of the ABI");
    if (!out || (!d_records && count) || nranks > kMaxCheckRanks || (nranks && (!rank_counts || !span_begin || !span_end))) return HAVAC_E_ARGUMENT;
    std::vector<uint64_t> table(3 * (size_t)nranks + 1, 0);      // rank_begin[nranks + 1], span_begin[nranks], span_end[nranks]
    for (uint32_t r = 0; r < nranks; r++) {
        table[r + 1] = table[r] + rank_counts[r];
        table[nranks + 1 + r] = span_begin[r];
        table[2 * (size_t)nranks + 1 + r] = span_end[r];
    }
    if (nranks && table[nranks] != count) return HAVAC_E_LENGTH;      // the ranks' counts do not add up to the list
    const hipStream_t stream = (hipStream_t)hip_stream;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HAVAC_E_NO_DEVICE;
    uint64_t* d_table = nullptr; OrderReport* d_report = nullptr;
    OrderReport report{count, 0, ~0ull, 0, ~0ull};
    hipError_t e = hipMalloc(&d_table, table.size() * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d_report, sizeof(OrderReport));
    if (e == hipSuccess) e = hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_report, &report, sizeof report, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && count) {
        // grid-stride: a dispatch counts its work-items in 32 bits, the list may hold more records than that
        const unsigned blocks = (unsigned)std::min<uint64_t>((count + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(ssv_check_order, dim3(blocks), dim3(256), 0, stream, d_records, count, (const uint64_t*)d_table,
                           (const uint64_t*)(d_table + nranks + 1), (const uint64_t*)(d_table + 2 * (size_t)nranks + 1), nranks, d_report);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&report, d_report, sizeof report, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (d_table) (void)hipFree(d_table);
    if (d_report) (void)hipFree(d_report);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? HAVAC_E_NOMEM : HAVAC_E_RUNTIME;
    out->records = report.records; out->out_of_order = report.out_of_order; out->first_out_of_order = report.first_out_of_order;
    out->out_of_span = report.out_of_span; out->first_out_of_span = report.first_out_of_span;
    return HAVAC_OK;
}

extern "C" int havac_ssv_last_ms(havac_ssv_ctx* c, float* ssv_kernel_ms, float* total_ms) {
    if (!c) return HAVAC_E_ARGUMENT;
    if (ssv_kernel_ms) *ssv_kernel_ms = c->ssv_ms;
    if (total_ms) *total_ms = c->total_ms;
    return HAVAC_OK;
}

// ===========================================================================
// Level 1: handle API (class HavacHwClient)
// ===========================================================================
// One handle drives one GPU (havac_dev_create) or several (havac_dev_create_multi): every GPU of the handle holds the
// whole sequence and model and computes one column shard (see shard_columns); because shards are whole segments, the
// GPUs' ordered hit lists concatenated in shard order are the ordered whole.
struct DevicePart {
    int device = 0;
    hipStream_t stream = nullptr;       // uploads, packing kernels; a run starts behind what this stream holds
    havac_pipe* pipe = nullptr;         // the runs of this GPU: `depth` slots (context, hit buffer, ordering stream each; havac_pipe.hip)
    uint8_t* d_seq = nullptr; uint64_t seq_alloc = 0;
    int8_t* d_phmm = nullptr; uint64_t phmm_alloc = 0;
    uint8_t* d_mask = nullptr; uint64_t mask_alloc = 0;   // separator bitmap, optional
    uint64_t win_first = 0, win_columns = 0;              // d_seq / d_mask hold these columns only (0, 0: all); see upload_columns
    uint32_t* d_abort = nullptr;        // kMaxDepth device words the kernels poll (cache-bypassing loads), one per run in flight
    hipStream_t abort_stream = nullptr; // abort() writes the words from here while the kernels run
    // text staging of havac_dev_write_sequence_chars (allocated on first use)
    char* d_chars[2] = {nullptr, nullptr};
    hipEvent_t chars_copied[2] = {nullptr, nullptr}, chars_packed[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
};

// One run of the handle: havac_dev_run_async opens it, havac_dev_wait / _state / _abort finish it, havac_dev_num_hits / _read_hits
// read it, the next havac_dev_run_async (depth 1) or havac_dev_retire closes it.
struct Run {
    uint32_t abort_word = 0;            // which of the parts' abort words this run polls
    bool abort_requested = false, finished = false, aborted = false, failed = false, overflowed = false;
    uint64_t found = 0;                 // all GPUs
    std::vector<uint64_t> part_found;
    std::vector<const uint64_t*> part_records;     // each GPU's ordered list (device memory of its pipe's slot)
    float ssv_ms = 0.f, total_ms = 0.f;
    std::string err;
};

constexpr uint32_t kMaxDepth = 4;
struct havac_dev {
    std::vector<DevicePart> parts;
    uint64_t seq_bytes = 0, phmm_bytes = 0, mask_bytes = 0;
    uint64_t hit_capacity = 0;          // per GPU and run in flight
    uint32_t depth = 1;                 // runs in flight (havac_dev_set_pipeline_depth); 1 = the reference's one at a time
    int tuning[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    std::deque<Run> runs;               // open runs, oldest first: the entry points below speak of the OLDEST
    bool has_run = false;               // a run was started at some time (the reference's "run object was not initialized")
    uint64_t submitted = 0;
    std::string err;
};

// the reference's fixed hit buffer: 14 * 256 MiB = 3.5 GiB = 469,762,048 records (host/HavacHwClient.hpp:94)
static const uint64_t kDefaultHitCapacity = 14ull * 256ull * 1024ull * 1024ull / sizeof(uint64_t);

static bool any_unfinished(const havac_dev* d) {
    for (const Run& r : d->runs) if (!r.finished) return true;
    return false;
}

static int apply_tuning(havac_dev* d) {
    const int* v = d->tuning;
    for (DevicePart& p : d->parts)
        for (uint32_t k = 0; k < d->depth; k++) {
            havac_ssv_ctx* const ctx = havac_pipe_context(p.pipe, (int)k);
            if (int rc = havac_ssv_set_tuning(ctx, v[0], v[1], v[2], v[3])) { d->err = havac_ssv_ctx_last_error(ctx); return rc; }
            if (int rc = havac_ssv_set_split_tuning(ctx, v[4], v[5], v[6], v[7])) { d->err = havac_ssv_ctx_last_error(ctx); return rc; }
            if (int rc = havac_ssv_set_kernel_variant(ctx, v[8])) { d->err = havac_ssv_ctx_last_error(ctx); return rc; }
        }
    return HAVAC_OK;
}

// (re)makes every GPU's pipe: `depth` slots with a hit buffer of `cap` records each.  Finished runs' lists go with the old pipes.
static int dev_make_pipes(havac_dev* d, uint32_t depth, uint64_t cap) {
    d->runs.clear();
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        if (p.pipe) { havac_pipe_destroy(p.pipe); p.pipe = nullptr; }
        // depth 1: one run at a time, the ordering on the kernel's stream (the reference's behaviour); deeper: the library's rule
        if (int rc = havac_pipe_create(depth, cap, -1, &p.pipe)) { d->err = "could not allocate the hit buffers and contexts of " + std::to_string(depth) + " run(s) in flight"; return rc; }
    }
    d->depth = depth; d->hit_capacity = cap;
    return apply_tuning(d);
}

extern "C" int havac_dev_create_multi(const uint32_t* device_indices, uint32_t ndevices, havac_dev** out) {
    if (!out || !device_indices || ndevices == 0) return HAVAC_E_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HAVAC_E_NO_DEVICE;
    for (uint32_t i = 0; i < ndevices; i++)
        if ((int)device_indices[i] >= ndev) return HAVAC_E_NO_DEVICE;
    havac_dev* d = new (std::nothrow) havac_dev;
    if (!d) return HAVAC_E_NOMEM;
    d->parts.resize(ndevices);
    auto fail = [&](int code) { havac_dev_destroy(d); return code; };
    for (uint32_t i = 0; i < ndevices; i++) {
        DevicePart& p = d->parts[i];
        p.device = (int)device_indices[i];
        if (hipSetDevice(p.device) != hipSuccess) return fail(HAVAC_E_NO_DEVICE);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, p.device) != hipSuccess) return fail(HAVAC_E_NO_DEVICE);
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(HAVAC_E_NO_DEVICE);   // the code object is gfx950 only
        if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) return fail(HAVAC_E_RUNTIME);
        if (hipStreamCreateWithFlags(&p.abort_stream, hipStreamNonBlocking) != hipSuccess) return fail(HAVAC_E_RUNTIME);
        if (hipMalloc(&p.d_abort, kMaxDepth * sizeof(uint32_t)) != hipSuccess) return fail(HAVAC_E_NOMEM);
        if (hipMemset(p.d_abort, 0, kMaxDepth * sizeof(uint32_t)) != hipSuccess) return fail(HAVAC_E_RUNTIME);
    }
    if (dev_make_pipes(d, 1, kDefaultHitCapacity) != HAVAC_OK) return fail(HAVAC_E_NOMEM);
    *out = d;
    return HAVAC_OK;
}

extern "C" int havac_dev_create(uint32_t device_index, havac_dev** out) {
    return havac_dev_create_multi(&device_index, 1, out);
}

extern "C" void havac_dev_destroy(havac_dev* d) {
    if (!d) return;
    for (DevicePart& p : d->parts) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.pipe) havac_pipe_destroy(p.pipe);
        if (p.d_seq) (void)hipFree(p.d_seq);
        if (p.d_phmm) (void)hipFree(p.d_phmm);
        if (p.d_mask) (void)hipFree(p.d_mask);
        if (p.d_abort) (void)hipFree(p.d_abort);
        for (int k = 0; k < 2; k++) {
            if (p.d_chars[k]) (void)hipFree(p.d_chars[k]);
            if (p.chars_copied[k]) (void)hipEventDestroy(p.chars_copied[k]);
            if (p.chars_packed[k]) (void)hipEventDestroy(p.chars_packed[k]);
        }
        if (p.copy_stream) (void)hipStreamDestroy(p.copy_stream);
        if (p.abort_stream) (void)hipStreamDestroy(p.abort_stream);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    delete d;
}

extern "C" const char* havac_dev_last_error(havac_dev* d) { return d ? d->err.c_str() : "null handle"; }

extern "C" uint32_t havac_dev_device_count(havac_dev* d) { return d ? (uint32_t)d->parts.size() : 0; }

extern "C" int havac_dev_set_hit_capacity(havac_dev* d, uint64_t max_hits) {
    if (!d || max_hits == 0) return HAVAC_E_ARGUMENT;
    if (any_unfinished(d)) { d->err = "cannot resize the hit buffer during a run"; return HAVAC_E_LOGIC; }
    return dev_make_pipes(d, d->depth, max_hits);
}

extern "C" int havac_dev_set_pipeline_depth(havac_dev* d, uint32_t depth) {
    if (!d || depth == 0 || depth > kMaxDepth) return HAVAC_E_ARGUMENT;
    if (any_unfinished(d)) { d->err = "cannot change the number of runs in flight during a run"; return HAVAC_E_LOGIC; }
    if (depth == d->depth) return HAVAC_OK;
    return dev_make_pipes(d, depth, d->hit_capacity);
}

extern "C" uint32_t havac_dev_pipeline_depth(havac_dev* d) { return d ? d->depth : 0; }

extern "C" int havac_dev_set_tuning(havac_dev* d, const int32_t* values, uint32_t count) {
    if (!d || (!values && count)) return HAVAC_E_ARGUMENT;
    if (any_unfinished(d)) { d->err = "cannot change the tuning during a run"; return HAVAC_E_LOGIC; }
    int v[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    for (uint32_t i = 0; i < count && i < 9; i++) v[i] = values[i];
    // everything is checked before anything is applied (the same tests the three setters make): a refused value leaves no GPU half tuned
    if (v[2] > 2 || v[3] > 1 || (v[0] > 0 && v[0] < 1024) || v[4] > 3 || (v[6] >= 0 && v[6] < 1024) || v[7] == 0 || v[7] == 1 || v[7] > 16 || v[8] > 1) {
        d->err = "bad tuning value";
        return HAVAC_E_ARGUMENT;
    }
    for (int i = 0; i < 9; i++) d->tuning[i] = v[i];
    return apply_tuning(d);
}

// Inputs may be rewritten while runs are in flight (depth > 1: the caller loads the next model while the last run still orders
// its records): the write waits until every run in flight has READ what is about to be overwritten -- the model is read by a
// pass's first kernel, the sequence by its SSV kernel -- not until the runs have finished.
static int wait_inputs_read(havac_dev* d, bool sequence_too) {
    if (!any_unfinished(d)) return HAVAC_OK;
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        if (int rc = havac_pipe_wait_inputs(p.pipe, sequence_too ? 1 : 0)) { d->err = havac_pipe_last_error(p.pipe); return rc; }
    }
    return HAVAC_OK;
}

// copies a host buffer to the same-named device buffer of every GPU of the handle
// Several GPUs behind the handle: the caller's pages are locked for the duration of the call, so that the per-GPU copies are
// truly asynchronous and cross PCIe side by side (from pageable memory every hipMemcpyAsync is staged by the runtime and the
// GPUs load one after the other: invisible at C4's 31 MB per GPU, eight times the time at the reference's 4 GiB limit).
// One GPU: nothing to overlap, nothing is locked.  If the pages cannot be locked the copies still work, staged.
struct LockedSource {
    void* p = nullptr;
    const havac_dev* dev;
    LockedSource(const havac_dev* d, const void* src, uint64_t nbytes) : dev(d) {
        if (d->parts.size() > 1 && nbytes >= (1u << 20)) {
            if (hipHostRegister(const_cast<void*>(src), nbytes, hipHostRegisterDefault) == hipSuccess) p = const_cast<void*>(src);
            else (void)hipGetLastError();
        }
    }
    ~LockedSource() {
        if (!p) return;
        // on an early error return some GPUs' copies from these pages may still be in flight: drained before the pages are unlocked
        for (const DevicePart& part : dev->parts)
            if (part.stream && hipSetDevice(part.device) == hipSuccess) (void)hipStreamSynchronize(part.stream);
        (void)hipHostUnregister(p);
    }
};
template <typename T>
static int upload(havac_dev* d, T* DevicePart::*buf, uint64_t DevicePart::*alloc, const void* src, uint64_t nbytes) {
    const LockedSource locked(d, src, nbytes);
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        if (p.*alloc < nbytes) {
            if (p.*buf) (void)hipFree(p.*buf);
            p.*buf = nullptr; p.*alloc = 0;
            HIP_TRY(d->err, hipMalloc(&(p.*buf), nbytes));
            p.*alloc = nbytes;
        }
        HIP_TRY(d->err, hipMemcpyAsync(p.*buf, src, nbytes, hipMemcpyHostToDevice, p.stream));
    }
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        HIP_TRY(d->err, hipStreamSynchronize(p.stream));
    }
    return HAVAC_OK;
}

// The same for a per-column buffer (the packed sequence: 4 columns per byte; the separator bitmap: 16) when the handle
// drives several GPUs: GPU i receives only the columns shard i can ever read -- its own and a left halo for the tallest
// model the record format allows (2^24 - 1 rows: 4 MB of packed sequence), since the model may be written later.  The
// reference has one device per object and a 4 GiB limit per object; eight GPUs each holding a full copy would be
// 8 x 4 GiB at that limit.
static const uint32_t kTallestModel = (1u << 24) - 1;
static void part_window(const havac_dev* d, uint64_t ncolumns, uint32_t part, uint64_t* first, uint64_t* columns) {
    const uint32_t nparts = (uint32_t)d->parts.size();
    if (nparts == 1 || ncolumns / HAVAC_SEGMENT_COLUMNS < nparts) { *first = 0; *columns = 0; return; }
    uint64_t b, e, wf, we;
    shard_columns(ncolumns, part, nparts, &b, &e);
    shard_window(ncolumns, kTallestModel, b, e, &wf, &we);
    *first = wf; *columns = we - wf;
}
template <typename T>
static int upload_columns(havac_dev* d, T* DevicePart::*buf, uint64_t DevicePart::*alloc, const uint8_t* src, uint64_t ncolumns,
                          uint32_t columns_per_byte) {
    const LockedSource locked(d, src, ncolumns / columns_per_byte);
    for (uint32_t i = 0; i < d->parts.size(); i++) {
        DevicePart& p = d->parts[i];
        HIP_TRY(d->err, hipSetDevice(p.device));
        part_window(d, ncolumns, i, &p.win_first, &p.win_columns);
        const uint64_t first = p.win_first, columns = p.win_columns ? p.win_columns : ncolumns;
        const uint64_t nbytes = columns / columns_per_byte;
        if (p.*alloc < nbytes) {
            if (p.*buf) (void)hipFree(p.*buf);
            p.*buf = nullptr; p.*alloc = 0;
            HIP_TRY(d->err, hipMalloc(&(p.*buf), nbytes));
            p.*alloc = nbytes;
        }
        HIP_TRY(d->err, hipMemcpyAsync(p.*buf, src + first / columns_per_byte, nbytes, hipMemcpyHostToDevice, p.stream));
    }
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        HIP_TRY(d->err, hipStreamSynchronize(p.stream));
    }
    return HAVAC_OK;
}
static void whole_copies(havac_dev* d) { for (DevicePart& p : d->parts) p.win_first = p.win_columns = 0; }

extern "C" int havac_dev_write_sequence(havac_dev* d, const uint8_t* packed, uint64_t nbytes) {
    if (!d || (!packed && nbytes)) return HAVAC_E_ARGUMENT;
    // host/HavacHwClient.cpp:81-97 (the reference computes the symbol count in 32 bits; we do not wrap)
    uint64_t nsym = nbytes * 4;
    if (nsym % HAVAC_SEGMENT_COLUMNS != 0) {
        d->err = "sequence length must be a multiple of the sequence segment length 12288 but sequence of length " +
                 std::to_string(nsym) + " was not \n";
        return HAVAC_E_LENGTH;
    }
    if (nbytes >= (4ull << 30)) {
        d->err = "compressed sequence size must be less than 4GiB. length requested: " + std::to_string(nbytes) + " bytes.";
        return HAVAC_E_LENGTH;
    }
    if (int rc = wait_inputs_read(d, true)) return rc;
    d->seq_bytes = nbytes;
    d->mask_bytes = 0;                       // a new sequence has no separators until a mask is written for it
    if (nbytes == 0) return HAVAC_OK;
    return upload_columns(d, &DevicePart::d_seq, &DevicePart::seq_alloc, packed, nbytes * 4, 4);
}

// SURVEY.md section 8 row f4.  Text in, packed on the GPU.  Per GPU: two staging buffers of kCharChunk characters;
// chunk c is copied on the copy stream while the pack kernel of chunk c-1 runs on the compute stream, and a
// staging buffer is reused only after the kernel that read it has finished.
static const uint64_t kCharChunk = 32ull << 20;

extern "C" int havac_dev_write_sequence_chars(havac_dev* d, const char* chars, uint64_t nchars, const uint64_t* patch_columns,
                                              const uint8_t* patch_symbols, uint64_t npatches) {
    if (!d || (!chars && nchars) || (npatches && (!patch_columns || !patch_symbols))) return HAVAC_E_ARGUMENT;
    const uint64_t ncolumns = (nchars + HAVAC_SEGMENT_COLUMNS - 1) / HAVAC_SEGMENT_COLUMNS * HAVAC_SEGMENT_COLUMNS;
    const uint64_t nbytes = ncolumns / 4;
    if (nbytes >= (4ull << 30)) {
        d->err = "compressed sequence size must be less than 4GiB. length requested: " + std::to_string(nbytes) + " bytes.";
        return HAVAC_E_LENGTH;
    }
    for (uint64_t i = 0; i < npatches; i++)
        if (patch_columns[i] >= ncolumns || (i && patch_columns[i] <= patch_columns[i - 1])) {
            d->err = "patch columns must be ascending and inside the padded sequence";
            return HAVAC_E_ARGUMENT;
        }
    if (int rc = wait_inputs_read(d, true)) return rc;
    d->seq_bytes = nbytes;
    d->mask_bytes = 0;
    whole_copies(d);
    if (nbytes == 0) return HAVAC_OK;
    // page-locked source: the chunk copies then run at PCIe speed and truly asynchronously; if the pages cannot be
    // locked the copies still work, through the runtime's own staging
    const bool locked = hipHostRegister(const_cast<char*>(chars), nchars, hipHostRegisterDefault) == hipSuccess;
    if (!locked) (void)hipGetLastError();
    int rc = HAVAC_OK;
    auto run = [&]() -> int {
        for (DevicePart& p : d->parts) {
            HIP_TRY(d->err, hipSetDevice(p.device));
            if (p.seq_alloc < nbytes) {
                if (p.d_seq) (void)hipFree(p.d_seq);
                p.d_seq = nullptr; p.seq_alloc = 0;
                HIP_TRY(d->err, hipMalloc(&p.d_seq, nbytes));
                p.seq_alloc = nbytes;
            }
            if (!p.d_chars[0]) {
                for (int k = 0; k < 2; k++) {
                    HIP_TRY(d->err, hipMalloc(&p.d_chars[k], kCharChunk));
                    HIP_TRY(d->err, hipEventCreateWithFlags(&p.chars_copied[k], hipEventDisableTiming));
                    HIP_TRY(d->err, hipEventCreateWithFlags(&p.chars_packed[k], hipEventDisableTiming));
                }
                HIP_TRY(d->err, hipStreamCreateWithFlags(&p.copy_stream, hipStreamNonBlocking));
            }
            uint32_t* const packed = reinterpret_cast<uint32_t*>(p.d_seq);
            uint64_t chunk = 0;
            for (uint64_t off = 0; off < ncolumns; off += kCharChunk, chunk++) {
                const int k = (int)(chunk & 1);
                const uint64_t have = off < nchars ? std::min(kCharChunk, nchars - off) : 0;     // characters in this chunk
                const uint64_t upto = std::min(off + kCharChunk, ncolumns);                      // columns it covers
                if (chunk >= 2) HIP_TRY(d->err, hipStreamWaitEvent(p.copy_stream, p.chars_packed[k], 0));
                if (have) HIP_TRY(d->err, hipMemcpyAsync(p.d_chars[k], chars + off, have, hipMemcpyHostToDevice, p.copy_stream));
                HIP_TRY(d->err, hipEventRecord(p.chars_copied[k], p.copy_stream));
                HIP_TRY(d->err, hipStreamWaitEvent(p.stream, p.chars_copied[k], 0));
                const uint64_t words = (upto - off) / 16;
                hipLaunchKernelGGL(ssv_pack_chars, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, p.stream,
                                   reinterpret_cast<const uint8_t*>(p.d_chars[k]), off, nchars, upto, packed);
                HIP_TRY(d->err, hipEventRecord(p.chars_packed[k], p.stream));
            }
            if (npatches) {
                uint64_t* d_cols = nullptr; uint8_t* d_syms = nullptr;
                HIP_TRY(d->err, hipMalloc(&d_cols, npatches * sizeof(uint64_t)));
                if (hipMalloc(&d_syms, npatches) != hipSuccess) { (void)hipFree(d_cols); d->err = "out of device memory"; return HAVAC_E_NOMEM; }
                hipError_t e = hipMemcpyAsync(d_cols, patch_columns, npatches * sizeof(uint64_t), hipMemcpyHostToDevice, p.stream);
                if (e == hipSuccess) e = hipMemcpyAsync(d_syms, patch_symbols, npatches, hipMemcpyHostToDevice, p.stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(ssv_patch_symbols, dim3((unsigned)((npatches + 255) / 256)), dim3(256), 0, p.stream,
                                       d_cols, d_syms, npatches, packed);
                    e = hipStreamSynchronize(p.stream);
                }
                (void)hipFree(d_cols); (void)hipFree(d_syms);
                if (e != hipSuccess) { d->err = hip_msg("patching the sequence", e); return HAVAC_E_RUNTIME; }
            }
        }
        for (DevicePart& p : d->parts) {
            HIP_TRY(d->err, hipSetDevice(p.device));
            HIP_TRY(d->err, hipStreamSynchronize(p.stream));
            HIP_TRY(d->err, hipGetLastError());
        }
        return HAVAC_OK;
    };
    rc = run();
    if (locked) (void)hipHostUnregister(const_cast<char*>(chars));
    return rc;
}

// the first `nbytes` of a per-column buffer as the handle's GPUs hold it: every GPU contributes its own columns
template <typename T>
static int read_columns(havac_dev* d, T* DevicePart::*buf, uint8_t* out, uint64_t nbytes, uint64_t ncolumns, uint32_t columns_per_byte) {
    const uint32_t nparts = (uint32_t)d->parts.size();
    for (uint32_t i = 0; i < nparts; i++) {
        DevicePart& p = d->parts[i];
        uint64_t b = 0, e = ncolumns;
        if (p.win_columns) {
            shard_columns(ncolumns, i, nparts, &b, &e);
            // shards are whole segments, and every writer refuses or pads a sequence that is not (havac_dev_write_sequence:
            // HAVAC_E_LENGTH); should a buffer ever end behind its last whole segment, the last GPU's window covers that tail
            // and returns it rather than leaving the caller's bytes unwritten
            if (i + 1 == nparts) e = std::min(ncolumns, p.win_first + p.win_columns);
        } else if (i > 0) break;                                 // whole copies: the first GPU has everything
        const uint64_t from = b / columns_per_byte, to = std::min(e / columns_per_byte, nbytes);
        if (from >= to) continue;
        HIP_TRY(d->err, hipSetDevice(p.device));
        HIP_TRY(d->err, hipMemcpy(out + from, p.*buf + (from - p.win_first / columns_per_byte), to - from, hipMemcpyDeviceToHost));
    }
    return HAVAC_OK;
}

extern "C" int havac_dev_read_sequence(havac_dev* d, uint8_t* out, uint64_t nbytes) {
    if (!d || (!out && nbytes)) return HAVAC_E_ARGUMENT;
    if (nbytes > d->seq_bytes) { d->err = "the device holds fewer sequence bytes than requested"; return HAVAC_E_LENGTH; }
    if (nbytes == 0) return HAVAC_OK;
    return read_columns(d, &DevicePart::d_seq, out, nbytes, d->seq_bytes * 4, 4);
}

extern "C" int havac_dev_read_separator_mask(havac_dev* d, uint8_t* out, uint64_t nbytes) {
    if (!d || (!out && nbytes)) return HAVAC_E_ARGUMENT;
    if (nbytes > d->mask_bytes) { d->err = "the device holds fewer separator-bitmap bytes than requested"; return HAVAC_E_LENGTH; }
    if (nbytes == 0) return HAVAC_OK;
    return read_columns(d, &DevicePart::d_mask, out, nbytes, d->seq_bytes * 4, 16);
}

namespace {
// host array -> a device array that lives until the caller frees it
template <typename T>
int to_device(std::string& err, const T* src, size_t n, T** out, hipStream_t stream) {
    *out = nullptr;
    if (n == 0) return HAVAC_OK;
    HIP_TRY(err, hipMalloc(out, n * sizeof(T)));
    hipError_t e = hipMemcpyAsync(*out, src, n * sizeof(T), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { (void)hipFree(*out); *out = nullptr; err = hip_msg("hipMemcpyAsync", e); return HAVAC_E_RUNTIME; }
    return HAVAC_OK;
}
}  // namespace

extern "C" int havac_dev_write_sequence_records(havac_dev* d, const char* chars, uint64_t nchars, const uint64_t* record_ends,
                                                uint32_t nrecords, uint64_t* record_starts_out) {
    if (!d || (!chars && nchars) || (nrecords && !record_ends)) return HAVAC_E_ARGUMENT;
    // the layout: record k occupies [start_k, start_k + len_k), then padding to an even column, then 2 separator columns
    std::vector<uint64_t> starts(nrecords), begins(nrecords), lens(nrecords);
    uint64_t at = 0;
    for (uint32_t k = 0; k < nrecords; k++) {
        const uint64_t begin = k ? record_ends[k - 1] : 0;
        if (record_ends[k] < begin || record_ends[k] > nchars) { d->err = "record ends must be ascending and inside the text"; return HAVAC_E_ARGUMENT; }
        starts[k] = at; begins[k] = begin; lens[k] = record_ends[k] - begin;
        at += lens[k];
        at += at & 1;
        at += 2;
    }
    if (record_starts_out) std::copy(starts.begin(), starts.end(), record_starts_out);
    const uint64_t ncolumns = (at + HAVAC_SEGMENT_COLUMNS - 1) / HAVAC_SEGMENT_COLUMNS * HAVAC_SEGMENT_COLUMNS;
    const uint64_t nbytes = ncolumns / 4;
    if (nbytes >= (4ull << 30)) {
        d->err = "compressed sequence size must be less than 4GiB. length requested: " + std::to_string(nbytes) + " bytes.";
        return HAVAC_E_LENGTH;
    }
    if (int rc = wait_inputs_read(d, true)) return rc;
    d->seq_bytes = nbytes;
    d->mask_bytes = 0;
    whole_copies(d);
    if (nbytes == 0) return HAVAC_OK;
    const bool locked = nchars && hipHostRegister(const_cast<char*>(chars), nchars, hipHostRegisterDefault) == hipSuccess;
    if (!locked) (void)hipGetLastError();
    auto run = [&]() -> int {
        for (DevicePart& p : d->parts) {
            HIP_TRY(d->err, hipSetDevice(p.device));
            if (p.seq_alloc < nbytes) {
                if (p.d_seq) (void)hipFree(p.d_seq);
                p.d_seq = nullptr; p.seq_alloc = 0;
                HIP_TRY(d->err, hipMalloc(&p.d_seq, nbytes));
                p.seq_alloc = nbytes;
            }
            if (p.mask_alloc < nbytes / 4) {
                if (p.d_mask) (void)hipFree(p.d_mask);
                p.d_mask = nullptr; p.mask_alloc = 0;
                HIP_TRY(d->err, hipMalloc(&p.d_mask, nbytes / 4));
                p.mask_alloc = nbytes / 4;
            }
            uint8_t* d_text = nullptr;
            uint64_t *d_starts = nullptr, *d_begins = nullptr, *d_lens = nullptr;
            auto release = [&]() { (void)hipFree(d_text); (void)hipFree(d_starts); (void)hipFree(d_begins); (void)hipFree(d_lens); };
            int rc = to_device(d->err, reinterpret_cast<const uint8_t*>(chars), (size_t)nchars, &d_text, p.stream);
            if (rc == HAVAC_OK) rc = to_device(d->err, starts.data(), starts.size(), &d_starts, p.stream);
            if (rc == HAVAC_OK) rc = to_device(d->err, begins.data(), begins.size(), &d_begins, p.stream);
            if (rc == HAVAC_OK) rc = to_device(d->err, lens.data(), lens.size(), &d_lens, p.stream);
            if (rc != HAVAC_OK) { release(); return rc; }
            const uint64_t words = ncolumns / 16;
            hipLaunchKernelGGL(ssv_pack_records, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, p.stream,
                               d_text, d_starts, d_begins, d_lens, nrecords, at, ncolumns,
                               reinterpret_cast<uint32_t*>(p.d_seq), p.d_mask);
            hipError_t e = hipStreamSynchronize(p.stream);
            if (e == hipSuccess) e = hipGetLastError();
            release();
            if (e != hipSuccess) { d->err = hip_msg("packing the records", e); return HAVAC_E_RUNTIME; }
        }
        return HAVAC_OK;
    };
    const int rc = run();
    if (locked) (void)hipHostUnregister(const_cast<char*>(chars));
    if (rc == HAVAC_OK) d->mask_bytes = nbytes / 4;
    return rc;
}

extern "C" int havac_dev_append_reverse_strand(havac_dev* d, const uint64_t* starts, const uint64_t* residues, uint32_t nrecords,
                                               uint64_t* forward_columns_out) {
    if (!d || (nrecords && (!starts || !residues))) return HAVAC_E_ARGUMENT;
    if (d->seq_bytes == 0) { d->err = "no sequence on the device to append a second strand to"; return HAVAC_E_LOGIC; }
    if (int rc = wait_inputs_read(d, true)) return rc;
    if (d->parts[0].win_columns) {
        d->err = "the GPUs of this handle hold column windows of a host-packed sequence (havac_dev_write_sequence): append the "
                 "second strand on the host, or send the text (havac_dev_write_sequence_chars / _records)";
        return HAVAC_E_LOGIC;
    }
    const uint64_t fbytes = d->seq_bytes, nf = fbytes * 4;
    if (2 * fbytes >= (4ull << 30)) {
        d->err = "compressed sequence size must be less than 4GiB. length requested: " + std::to_string(2 * fbytes) + " bytes.";
        return HAVAC_E_LENGTH;
    }
    for (uint32_t k = 0; k < nrecords; k++)
        if (starts[k] + residues[k] > nf || (k && starts[k] < starts[k - 1] + residues[k - 1])) {
            d->err = "records must be ascending, disjoint and inside the forward sequence";
            return HAVAC_E_ARGUMENT;
        }
    for (DevicePart& p : d->parts) {
        HIP_TRY(d->err, hipSetDevice(p.device));
        if (p.seq_alloc < 2 * fbytes) {          // grow, keeping the forward half
            uint8_t* bigger = nullptr;
            HIP_TRY(d->err, hipMalloc(&bigger, 2 * fbytes));
            hipError_t e = hipMemcpyAsync(bigger, p.d_seq, fbytes, hipMemcpyDeviceToDevice, p.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
            if (e != hipSuccess) { (void)hipFree(bigger); d->err = hip_msg("growing the sequence buffer", e); return HAVAC_E_RUNTIME; }
            (void)hipFree(p.d_seq);
            p.d_seq = bigger; p.seq_alloc = 2 * fbytes;
        }
        uint64_t *d_starts = nullptr, *d_residues = nullptr;
        int rc = to_device(d->err, starts, nrecords, &d_starts, p.stream);
        if (rc == HAVAC_OK) rc = to_device(d->err, residues, nrecords, &d_residues, p.stream);
        if (rc != HAVAC_OK) { (void)hipFree(d_starts); (void)hipFree(d_residues); return rc; }
        const uint64_t words = nf / 16;
        hipLaunchKernelGGL(ssv_reverse_strand, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, p.stream,
                           reinterpret_cast<uint32_t*>(p.d_seq), nf, d_starts, d_residues, nrecords);
        hipError_t e = hipSuccess;
        if (d->mask_bytes) {                        // the second half has the separators of the first
            if (p.mask_alloc < 2 * d->mask_bytes) {
                uint8_t* bigger = nullptr;
                e = hipMalloc(&bigger, 2 * d->mask_bytes);
                if (e == hipSuccess) e = hipMemcpyAsync(bigger, p.d_mask, d->mask_bytes, hipMemcpyDeviceToDevice, p.stream);
                if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
                if (e == hipSuccess) { (void)hipFree(p.d_mask); p.d_mask = bigger; p.mask_alloc = 2 * d->mask_bytes; }
                else if (bigger) (void)hipFree(bigger);
            }
            if (e == hipSuccess) e = hipMemcpyAsync(p.d_mask + d->mask_bytes, p.d_mask, d->mask_bytes, hipMemcpyDeviceToDevice, p.stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
        if (e == hipSuccess) e = hipGetLastError();
        (void)hipFree(d_starts); (void)hipFree(d_residues);
        if (e != hipSuccess) { d->err = hip_msg("appending the second strand", e); return HAVAC_E_RUNTIME; }
    }
    d->seq_bytes = 2 * fbytes;
    d->mask_bytes *= 2;
    if (forward_columns_out) *forward_columns_out = nf;
    return HAVAC_OK;
}

extern "C" int havac_dev_write_separator_mask(havac_dev* d, const uint8_t* pair_bitmap, uint64_t nbytes) {
    if (!d || (!pair_bitmap && nbytes)) return HAVAC_E_ARGUMENT;
    if (int rc = wait_inputs_read(d, true)) return rc;
    if (nbytes == 0) { d->mask_bytes = 0; return HAVAC_OK; }
    if (nbytes != d->seq_bytes / 4) {        // one bit per symbol pair = 1/8 bit per packed bit
        d->err = "separator mask must hold one bit per symbol pair of the sequence written before it (" +
                 std::to_string(d->seq_bytes / 4) + " bytes), got " + std::to_string(nbytes);
        return HAVAC_E_LENGTH;
    }
    d->mask_bytes = nbytes;
    // (the same columns per GPU as the sequence it belongs to)
    if (d->parts[0].win_columns == 0) return upload(d, &DevicePart::d_mask, &DevicePart::mask_alloc, pair_bitmap, nbytes);
    return upload_columns(d, &DevicePart::d_mask, &DevicePart::mask_alloc, pair_bitmap, d->seq_bytes * 4, 16);
}

extern "C" int havac_dev_write_phmm(havac_dev* d, const int8_t* scores, uint64_t nbytes) {
    if (!d || (!scores && nbytes)) return HAVAC_E_ARGUMENT;
    // host/HavacHwClient.cpp:112-125
    if (nbytes % 4 != 0) {
        d->err = "phmm of length length " + std::to_string(nbytes) +
                 " was given, but was not divisible by 4. only nucleotide phmms with 4 scores/position is supported.\n";
        return HAVAC_E_LENGTH;
    }
    if (nbytes >= (1ull << 30)) {
        d->err = "model length must be less than 1MiB (model length < 1024*1024). length requested: " +
                 std::to_string(nbytes) + " bytes.";
        return HAVAC_E_LENGTH;
    }
    if (int rc = wait_inputs_read(d, false)) return rc;
    d->phmm_bytes = nbytes;
    if (nbytes == 0) return HAVAC_OK;
    return upload(d, &DevicePart::d_phmm, &DevicePart::phmm_alloc, scores, nbytes);
}

// ---- runs ----------------------------------------------------------------------------------------------------------------
// Up to `depth` runs are open at a time (havac_dev_set_pipeline_depth; 1 = the reference's one at a time).  Every entry point
// below speaks of the OLDEST open run; havac_dev_run_async opens a new one -- at depth 1 a finished run is closed by the next
// run_async, as the reference's run object is overwritten by the next invokeHavacSsvAsync (host/HavacHwClient.cpp:149); with
// more runs in flight havac_dev_retire closes the oldest and makes the next one current.
static Run* first_unfinished(havac_dev* d);
static int dev_finish(havac_dev* d, Run& r);
extern "C" int havac_dev_run_async(havac_dev* d) {
    if (!d) return HAVAC_E_ARGUMENT;
    // host/HavacHwClient.cpp:142-147
    if (d->seq_bytes == 0) { d->err = "sequence length in segments cannot be 0, but 0 was given to the client."; return HAVAC_E_LENGTH; }
    if (d->phmm_bytes == 0) { d->err = "phmm length in vectors cannot be 0, but 0 was given to the client."; return HAVAC_E_LENGTH; }
    if (d->runs.size() == d->depth) {
        if (!d->runs.front().finished) { d->err = d->depth == 1 ? "a run is already in flight" : "every run slot is in flight: wait for the oldest run and retire it (havac_dev_retire)"; return HAVAC_E_LOGIC; }
        d->runs.pop_front();            // a finished run makes room (its list goes with it)
    }
    const uint32_t nparts = (uint32_t)d->parts.size();
    if (d->seq_bytes * 4 / HAVAC_SEGMENT_COLUMNS < nparts) {
        d->err = "the sequence has fewer 12288-column segments than the handle has GPUs";
        return HAVAC_E_LENGTH;
    }
    Run run;
    run.abort_word = (uint32_t)(d->submitted % kMaxDepth);
    run.part_found.assign(nparts, 0); run.part_records.assign(nparts, nullptr);
    for (uint32_t i = 0; i < nparts; i++) {
        DevicePart& p = d->parts[i];
        HIP_TRY(d->err, hipSetDevice(p.device));
        havac_ssv_ctx* const ctx = havac_pipe_context(p.pipe, -2);          // the slot this run will take
        havac_ssv_set_separator_mask(ctx, d->mask_bytes ? p.d_mask : nullptr);
        havac_ssv_set_sequence_window(ctx, p.win_first, p.win_columns);
        // (nothing to wait for: every write_* entry point returns with its upload complete; and the abort words are zero -- the run
        // that was aborted last put its own back, see dev_finish -- so a run costs no fill and no event in front of its first kernel)
        int rc = havac_pipe_submit(p.pipe, p.d_seq, d->seq_bytes * 4, p.d_phmm, (uint32_t)(d->phmm_bytes / 4), i, nparts,
                                   p.d_abort + run.abort_word, HAVAC_NO_STREAM);
        if (rc) {
            d->err = havac_pipe_last_error(p.pipe);
            // take back what the earlier GPUs were given: they are the newest pass of their pipes; every older open run is finished
            // first, so that the pipes stay in step (a refused submit is rare: an argument error, or out of memory)
            if (i > 0) {
                while (Run* older = first_unfinished(d)) (void)dev_finish(d, *older);
                for (uint32_t j = 0; j < i; j++) {
                    (void)hipSetDevice(d->parts[j].device);
                    (void)havac_pipe_collect(d->parts[j].pipe, nullptr, nullptr, nullptr, d->parts[j].stream);
                }
            }
            return rc;
        }
    }
    d->runs.push_back(std::move(run));
    d->submitted++;
    d->has_run = true;
    return HAVAC_OK;
}

static int final_state(const Run& r) {
    return r.failed ? HAVAC_STATE_ERROR : (r.aborted ? HAVAC_STATE_ABORT : HAVAC_STATE_COMPLETED);
}

// the first open run that has not been finished yet (runs finish in order: a pipe completes its oldest pass first)
static Run* first_unfinished(havac_dev* d) {
    for (Run& r : d->runs) if (!r.finished) return &r;
    return nullptr;
}

// completes run r, the oldest UNFINISHED one: every GPU's pass is collected (all of it was enqueued at run_async, the ordering
// included, so the GPUs have been working side by side; the waits here are one after the other and cost nothing extra)
static int dev_finish(havac_dev* d, Run& r) {
    if (r.finished) return final_state(r);
    r.finished = true;
    r.found = 0;
    for (size_t i = 0; i < d->parts.size(); i++) {
        DevicePart& p = d->parts[i];
        (void)hipSetDevice(p.device);
        uint64_t found = 0, n = 0;
        const uint64_t* records = nullptr;
        const int rc = havac_pipe_collect(p.pipe, &found, &records, &n, p.stream);
        r.part_found[i] = found; r.part_records[i] = records;
        r.found += found;
        float a = 0.f, b = 0.f;
        if (havac_pipe_last_ms(p.pipe, &a, &b) == HAVAC_OK) { r.ssv_ms = std::max(r.ssv_ms, a); r.total_ms = std::max(r.total_ms, b); }      // the slowest GPU
        if (rc != HAVAC_OK) {
            r.err = havac_pipe_last_error(p.pipe);
            d->err = r.err;
            r.failed = true;
            if (rc == HAVAC_E_HIT_OVERFLOW) r.overflowed = true;
        }
        if (r.abort_requested) {      // the word this run was stopped through is put back for the runs to come
            (void)hipMemsetAsync(p.d_abort + r.abort_word, 0, sizeof(uint32_t), p.abort_stream);
            (void)hipStreamSynchronize(p.abort_stream);
        }
    }
    r.aborted = r.abort_requested;
    return final_state(r);
}

// 1 when every GPU's pass of the oldest unfinished run has drained, 0 while any has not, negative on error
static int query_all(havac_dev* d) {
    for (DevicePart& p : d->parts) {
        const int q = havac_pipe_poll(p.pipe);
        if (q <= 0) { if (q < 0) d->err = havac_pipe_last_error(p.pipe); return q; }
    }
    return 1;
}

extern "C" int havac_dev_state(havac_dev* d) {
    if (!d) return HAVAC_E_ARGUMENT;
    if (!d->has_run || d->runs.empty()) {   // host/HavacHwClient.cpp:168
        d->err = "run object was not initialized. run function invokeHavacSsvAsync to initialize this object.";
        return HAVAC_E_LOGIC;
    }
    Run& cur = d->runs.front();
    if (cur.finished) return final_state(cur);
    const int q = query_all(d);
    if (q == 0) return HAVAC_STATE_RUNNING;
    if (q < 0) return HAVAC_STATE_ERROR;
    return dev_finish(d, cur);
}

extern "C" int havac_dev_wait(havac_dev* d, uint32_t timeout_ms) {
    if (!d) return HAVAC_E_ARGUMENT;
    if (!d->has_run || d->runs.empty()) { d->err = "no run to wait for"; return HAVAC_E_LOGIC; }
    Run& cur = d->runs.front();
    if (cur.finished) return final_state(cur);
    if (timeout_ms == 0) return dev_finish(d, cur);
    auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        const int q = query_all(d);
        if (q > 0) return dev_finish(d, cur);
        if (q < 0) { cur.finished = true; cur.failed = true; return HAVAC_STATE_ERROR; }
        if (std::chrono::steady_clock::now() >= deadline) return HAVAC_STATE_TIMEOUT;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

extern "C" int havac_dev_abort(havac_dev* d) {
    if (!d) return HAVAC_E_ARGUMENT;
    if (!d->has_run || d->runs.empty()) { d->err = "no run to abort"; return HAVAC_E_LOGIC; }
    Run& cur = d->runs.front();
    if (cur.finished) return final_state(cur);
    if (query_all(d) > 0) return dev_finish(d, cur);
    // every run in flight is stopped: the oldest is finished here, the others report ABORT when their turn comes
    static const uint32_t one = 1;
    for (Run& r : d->runs) {
        if (r.finished) continue;
        r.abort_requested = true;
        for (DevicePart& p : d->parts) {
            hipError_t e = hipSetDevice(p.device);
            if (e == hipSuccess) e = hipMemcpyAsync(p.d_abort + r.abort_word, &one, sizeof one, hipMemcpyHostToDevice, p.abort_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(p.abort_stream);
            if (e != hipSuccess) { d->err = hip_msg("abort", e); cur.finished = true; cur.failed = true; return HAVAC_STATE_ERROR; }
        }
    }
    return dev_finish(d, cur);
}

extern "C" int havac_dev_retire(havac_dev* d) {
    if (!d) return HAVAC_E_ARGUMENT;
    if (d->runs.empty()) { d->err = "no open run to retire"; return HAVAC_E_LOGIC; }
    if (!d->runs.front().finished) { const int s = havac_dev_wait(d, 0); if (s < 0) return s; }
    d->runs.pop_front();
    return HAVAC_OK;
}

extern "C" uint32_t havac_dev_open_runs(havac_dev* d) { return d ? (uint32_t)d->runs.size() : 0; }

extern "C" int havac_dev_num_hits64(havac_dev* d, uint64_t* count) {
    if (!d || !count) return HAVAC_E_ARGUMENT;
    if (!d->has_run || d->runs.empty()) { d->err = "num hits was not set by the client!"; return HAVAC_E_RUNTIME; }   // HavacHwClient.cpp:181-183
    Run& cur = d->runs.front();
    if (!cur.finished) { int s = havac_dev_wait(d, 0); if (s < 0) return s; }
    if (cur.failed) { d->err = cur.err; return cur.overflowed ? HAVAC_E_HIT_OVERFLOW : HAVAC_E_RUNTIME; }
    if (cur.aborted) {       // the sweep stopped part-way: whatever it had queued is not a hit list
        d->err = "the run was aborted: it has no hit list";
        return HAVAC_E_LOGIC;
    }
    *count = cur.found;
    return HAVAC_OK;
}

extern "C" int havac_dev_num_hits(havac_dev* d, uint32_t* count) {
    if (!count) return HAVAC_E_ARGUMENT;
    uint64_t found = 0;
    int rc = havac_dev_num_hits64(d, &found);
    if (rc) return rc;
    if (found > 0xffffffffull) {          // the reference's counter is 32 bits wide (host/HavacHwClient.cpp:172-186)
        d->err = std::to_string(found) + " hits do not fit the 32-bit count of this entry point: use havac_dev_num_hits64";
        return HAVAC_E_HIT_OVERFLOW;
    }
    *count = (uint32_t)found;
    return HAVAC_OK;
}

extern "C" int havac_dev_read_hits64(havac_dev* d, uint64_t* out, uint64_t n) {
    if (!d || (!out && n)) return HAVAC_E_ARGUMENT;
    uint64_t have = 0;
    int rc = havac_dev_num_hits64(d, &have);
    if (rc) return rc;
    if (n > have) n = have;
    const Run& cur = d->runs.front();
    // shard order is device order: the GPUs' lists are simply laid end to end -- every GPU's copy is enqueued before any is
    // waited for, so with several GPUs behind the handle the copies cross PCIe side by side.  (On the GPU's abort stream: the
    // handle's own stream may already hold the next run's uploads, and the list is complete -- the run was waited for.)
    uint64_t at = 0;
    std::vector<DevicePart*> copying;
    for (size_t i = 0; i < d->parts.size(); i++) {
        DevicePart& p = d->parts[i];
        if (at >= n) break;
        uint64_t take = cur.part_found[i] < n - at ? cur.part_found[i] : n - at;
        if (take == 0) continue;
        HIP_TRY(d->err, hipSetDevice(p.device));
        HIP_TRY(d->err, hipMemcpyAsync(out + at, cur.part_records[i], (size_t)take * sizeof(uint64_t), hipMemcpyDeviceToHost, p.abort_stream));
        copying.push_back(&p);
        at += take;
    }
    for (DevicePart* p : copying) {
        HIP_TRY(d->err, hipSetDevice(p->device));
        HIP_TRY(d->err, hipStreamSynchronize(p->abort_stream));
    }
    return HAVAC_OK;
}

extern "C" int havac_dev_read_hits(havac_dev* d, uint64_t* out, uint32_t n) { return havac_dev_read_hits64(d, out, n); }

extern "C" int havac_dev_last_run_ms(havac_dev* d, float* ssv_kernel_ms, float* total_ms) {
    if (!d) return HAVAC_E_ARGUMENT;
    if (!d->has_run || d->runs.empty() || !d->runs.front().finished) { d->err = "no finished run"; return HAVAC_E_LOGIC; }
    if (ssv_kernel_ms) *ssv_kernel_ms = d->runs.front().ssv_ms;
    if (total_ms) *total_ms = d->runs.front().total_ms;
    return HAVAC_OK;
}
