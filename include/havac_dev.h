/*
 * havac_dev.h -- C ABI of the MI355X device layer for HAVAC's SSV filter.
 *
 * This is the drop-in boundary: it replaces the reference's XRT wrapper
 * `HavacHwClient` (host/HavacHwClient.hpp:22-75, host/HavacHwClient.cpp:25-202)
 * and, below it, the Vitis-HLS kernel `HavacKernel` (device/HavacHls.cpp:20-43).
 * Plain pointers and sizes only; no C++, HIP or torch types cross it.  The
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Three levels are exported by libhavac_dev.so (the third -- the RCCL gather of a sharded run -- at the end of this file; between
 * the second and the third, "2b": several passes in flight):
 *
 *  1. havac_dev_*   one opaque handle per `Havac` object; one entry point per
 *                   HavacHwClient method the `Havac` class calls
 *                   (host/Havac.cpp:24,53,75,93,97,101,146,191).  The handle
 *                   owns its device buffers, as HavacHwClient owns its xrt::bo's.
 *  2. havac_ssv_*   the kernel ABI itself on caller-owned DEVICE memory and a
 *                   caller-owned HIP stream: the counterpart of calling
 *                   HavacKernel(seq, nSeg, phmm, sumL, hits, hitCount) directly
 *                   (device/HavacHls.cpp:20-43).  Used by bench.py and by the
 *                   multi-GPU driver, which keep inputs resident in HBM.
 *
 * Data contracts (identical to the reference's):
 *   sequence  2-bit packed, 4 symbols per byte, symbol i in byte i/4 at bits
 *             (i%4)*2, A=0 C=1 G=2 T=3 (host/sequence/SequencePreprocessor.cpp:46-70);
 *             symbol count a multiple of 12288 (host/HavacHwClient.cpp:81-90);
 *             fewer than 4 GiB of packed bytes (host/HavacHwClient.cpp:92-97).
 *   model     flattened int8 [row][A,C,G,T], all models of the .hmm file back to
 *             back (host/phmm/PhmmPreprocessor.cpp:9-31); byte count a multiple
 *             of 4 and below 1 GiB (host/HavacHwClient.cpp:112-125); rows < 2^24
 *             (device/HavacHls.hpp:19).
 *   hits      packed 64-bit records: [13:0] column inside its 12288-column
 *             segment, [39:14] segment index, [63:40] global model row
 *             (device/HitReporting.cpp:421-430; decoded at host/Havac.cpp:155-163),
 *             returned in the FPGA's emission order: segment ascending, then row
 *             ascending, then column ascending (device/HavacHls.cpp:151-152,264).
 *
 * Every function returns HAVAC_OK (0) or a negative HAVAC_E_* code; the text
 * of the last failure on a handle is available from havac_dev_last_error().
 * A handle is not thread-safe (neither is HavacHwClient).
 */
#ifndef HAVAC_DEV_H
#define HAVAC_DEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAVAC_SEGMENT_COLUMNS 12288u /* device/PublicDefines.h:18-22 */

/* error codes; the C++ `Havac` wrapper maps them back to the exception types
 * the reference throws (noted per code) */
enum {
    HAVAC_OK = 0,
    HAVAC_E_LENGTH = -1,       /* std::length_error  host/HavacHwClient.cpp:88,96,118,124,143,146 */
    HAVAC_E_LOGIC = -2,        /* std::logic_error   host/HavacHwClient.cpp:168, host/Havac.cpp:86-91 */
    HAVAC_E_RUNTIME = -3,      /* std::runtime_error host/HavacHwClient.cpp:179,182 and HIP failures */
    HAVAC_E_NOMEM = -4,        /* std::bad_alloc */
    HAVAC_E_HIT_OVERFLOW = -5, /* more hits than the hit buffer holds (the reference has a fixed 3.5 GiB
                                  buffer and no check, host/HavacHwClient.hpp:94) */
    HAVAC_E_NO_DEVICE = -6,    /* no usable gfx950 device / HIP runtime */
    HAVAC_E_ARGUMENT = -7,
    HAVAC_E_TIMEOUT = -8       /* a collective of a sharded run did not complete before its deadline (havac_gather_set_deadline);
                                  std::runtime_error.  No counterpart in the reference: one device per object */
};

/* run states, numerically equal to havac_cmd_state / ert_cmd_state
 * (host/Havac.hpp:16-26) */
enum {
    HAVAC_STATE_NEW = 1,
    HAVAC_STATE_QUEUED = 2,
    HAVAC_STATE_RUNNING = 3,
    HAVAC_STATE_COMPLETED = 4,
    HAVAC_STATE_ERROR = 5,
    HAVAC_STATE_ABORT = 6,
    HAVAC_STATE_SUBMITTED = 7,
    HAVAC_STATE_TIMEOUT = 8,
    HAVAC_STATE_NORESPONSE = 9
};

/* ------------------------------------------------------------------------
 * Level 1: handle API (replaces class HavacHwClient)
 * ---------------------------------------------------------------------- */
typedef struct havac_dev havac_dev;

/* HavacHwClient::HavacHwClient(xclbin, kernelName, deviceIndex)
 * host/HavacHwClient.cpp:25-31.  There is no bitstream to load; the kernels
 * are inside the library.  Allocates the hit buffer with the reference's
 * capacity, 14*256 MiB = 469,762,048 records (host/HavacHwClient.hpp:94);
 * see havac_dev_set_hit_capacity. */
int havac_dev_create(uint32_t device_index, havac_dev **out);
/* The same handle over several GPUs of one node (the reference has one deviceIndex per
 * object, host/Havac.hpp:51).  Every GPU receives the whole sequence and model; GPU i computes column shard i of
 * ndevices (see havac_ssv_enqueue); havac_dev_read_hits lays the GPUs' ordered lists end to end, which is the
 * reference's device order of the whole.  All other entry points behave as for one GPU; the hit capacity is per
 * GPU.  A device index may be repeated (several shards on one GPU). */
int havac_dev_create_multi(const uint32_t *device_indices, uint32_t ndevices, havac_dev **out);
uint32_t havac_dev_device_count(havac_dev *dev);
void havac_dev_destroy(havac_dev *dev);

/* Change the hit-buffer capacity (records). */
int havac_dev_set_hit_capacity(havac_dev *dev, uint64_t max_hits);
/* Experiment knobs of the handle's runs: up to nine values, in the order of havac_ssv_set_tuning's, then
 * havac_ssv_set_split_tuning's arguments, then havac_ssv_set_kernel_variant's (see there; missing values and -1 = the library's
 * own rule).  Applies to every
 * GPU of the handle.  No counterpart in the reference. */
int havac_dev_set_tuning(havac_dev *dev, const int32_t *values, uint32_t count);

/* HavacHwClient::writeSequence  host/HavacHwClient.cpp:78-110.
 * Borrowed host pointer, copied to HBM before return. */
int havac_dev_write_sequence(havac_dev *dev, const uint8_t *packed2bit, uint64_t nbytes);

/* Optional, not in the reference (packing on the GPU, pipelined host-to-device copy).
 * The sequence as TEXT: `chars` holds nchars bytes, one per column -- the FASTA residues with the '\0' that ends
 * every record, exactly what host/sequence/SequencePreprocessor.cpp:9-59 packs.  a/c/g/t (either case) are packed
 * to 2 bits on the GPU, chunk by chunk while later chunks are still crossing PCIe; every other column is listed in
 * `patch_columns` (ascending) with the symbol 0..3 the host drew for it in the reference's rand() order
 * (SequencePreprocessor.cpp:61-84), and is patched in afterwards.  Columns from nchars up to the next multiple of
 * 12288 are symbol 0, as the reference pads.  The device buffer ends up byte for byte what
 * havac_dev_write_sequence would have stored for the host-packed sequence. */
int havac_dev_write_sequence_chars(havac_dev *dev, const char *chars, uint64_t nchars, const uint64_t *patch_columns,
                                   const uint8_t *patch_symbols, uint64_t npatches);

/* The packed sequence as it stands in HBM (first GPU of the handle): for checking the entry points around it. */
int havac_dev_read_sequence(havac_dev *dev, uint8_t *packed2bit, uint64_t nbytes);
/* The separator bitmap as it stands in HBM (first GPU of the handle); nbytes <= sequence bytes / 4. */
int havac_dev_read_separator_mask(havac_dev *dev, uint8_t *pair_bitmap, uint64_t nbytes);

/* Optional, not in the reference (rows f2 + f4: the boundary-mode layout built on the GPU).  `chars` as for
 * havac_dev_write_sequence_chars; record k is chars[record_ends[k-1] .. record_ends[k]) (its residues and the '\0'
 * behind them; record_ends ascending, the last one == nchars).  Every record gets its own columns followed, at the
 * next even column, by one separator pair; a/c/g are 0/1/2 and every other character is T (host/test/Ssv.cpp:29-34,
 * no rand()); the sequence and its separator bitmap are both made on the GPU and equal, byte for byte, what the host
 * layer's SequencePreprocessor(fasta, true) packs.  record_starts_out (nrecords entries, may be NULL) receives each
 * record's first column. */
int havac_dev_write_sequence_records(havac_dev *dev, const char *chars, uint64_t nchars, const uint64_t *record_ends,
                                     uint32_t nrecords, uint64_t *record_starts_out);

/* Optional, not in the reference (row f3: both strands, made on the GPU).  Doubles the sequence that is on the device
 * (and its separator bitmap, if any): columns [Nf, 2 Nf) repeat [0, Nf) with the `residues[k]` columns from
 * `starts[k]` on reverse-complemented in place for every record k (starts ascending); everything else is copied.
 * *forward_columns_out = Nf.  Equals SequencePreprocessor::appendReverseStrand on the host-packed buffer. */
int havac_dev_append_reverse_strand(havac_dev *dev, const uint64_t *starts, const uint64_t *residues, uint32_t nrecords,
                                    uint64_t *forward_columns_out);

/* Optional, not in the reference ("boundary mode").  One bit per aligned symbol
 * pair (bit k of byte j = symbols 16j+2k, 16j+2k+1) of the sequence written before it; a set bit makes both
 * symbols score -128 on every model row, which resets every diagonal that crosses the pair (255-128-128 < 0)
 * and can never hit.  Used by the host layer to separate FASTA records.  nbytes must be sequence bytes / 4;
 * nbytes == 0 removes the mask.  Writing a new sequence removes it too. */
int havac_dev_write_separator_mask(havac_dev *dev, const uint8_t *pair_bitmap, uint64_t nbytes);

/* HavacHwClient::writePhmm  host/HavacHwClient.cpp:111-138. */
int havac_dev_write_phmm(havac_dev *dev, const int8_t *scores, uint64_t nbytes);

/* HavacHwClient::invokeHavacSsvAsync  host/HavacHwClient.cpp:141-151.
 * Enqueues on the handle's private stream and returns. */
int havac_dev_run_async(havac_dev *dev);

/* Optional, not in the reference, whose client runs one pass at a time (host/HavacHwClient.cpp:141-157): up to `depth` runs open at
 * once (1, the default, ... 4).  havac_dev_run_async may then be called again while earlier runs are still on the device -- the
 * records of run k are ordered and read back while the kernel of run k + 1 runs, which itself started while kernel k was draining
 * (level 2b below: every GPU of the handle gets a pipe of `depth` slots, a hit buffer of the handle's capacity each).  All other
 * entry points -- state, wait, abort, num_hits, read_hits, last_run_ms -- speak of the OLDEST open run; havac_dev_retire closes
 * it (waiting for it if need be) and makes the next one current.  Results are retrieved in submission order.  At depth 1
 * nothing changes: a finished run is closed by the next havac_dev_run_async, as the reference's run object is overwritten by
 * the next invokeHavacSsvAsync.  The inputs may be rewritten between runs while earlier runs are in flight: a write waits until
 * those runs have READ what it overwrites (the model: a pass's first kernel; the sequence: its SSV kernel), not until they have
 * finished.  havac_dev_abort stops every run in flight.  Changing the depth (or the hit capacity) needs every open run finished
 * and drops their lists. */
int havac_dev_set_pipeline_depth(havac_dev *dev, uint32_t depth);
uint32_t havac_dev_pipeline_depth(havac_dev *dev);
int havac_dev_retire(havac_dev *dev);
uint32_t havac_dev_open_runs(havac_dev *dev);

/* HavacHwClient::getHwState  host/HavacHwClient.cpp:163-170.
 * Returns a HAVAC_STATE_* value (>0) or a negative error (LOGIC when no run
 * was ever started, as the reference throws). */
int havac_dev_state(havac_dev *dev);

/* HavacHwClient::waitForHavacSsvAsync  host/HavacHwClient.cpp:153-157.
 * timeout_ms == 0 waits without limit (xrt::run::wait semantics).  Returns the
 * state on return: COMPLETED, TIMEOUT, ABORT or ERROR. */
int havac_dev_wait(havac_dev *dev, uint32_t timeout_ms);

/* HavacHwClient::abort  host/HavacHwClient.cpp:159-161.  Raises a flag the
 * kernel polls between row chunks, then waits for the stream to drain.
 * Returns HAVAC_STATE_ABORT, or COMPLETED if the run had already finished. */
int havac_dev_abort(havac_dev *dev);

/* HavacHwClient::getNumHits / getHitList  host/HavacHwClient.cpp:172-202.
 * Valid after a completed run (HAVAC_E_LOGIC after an aborted one: a sweep
 * that was stopped part-way has no hit list).  read_hits copies min(n, count)
 * records in device order. */
int havac_dev_num_hits(havac_dev *dev, uint32_t *count);
/* The same with 64-bit counts: several GPUs behind one handle can hold more than 2^32 - 1 records (C4: 4.5e9);
 * havac_dev_num_hits returns HAVAC_E_HIT_OVERFLOW then instead of a truncated count. */
int havac_dev_num_hits64(havac_dev *dev, uint64_t *count);
int havac_dev_read_hits64(havac_dev *dev, uint64_t *out, uint64_t n);
int havac_dev_read_hits(havac_dev *dev, uint64_t *out, uint32_t n);

/* Device time of the last completed run in milliseconds (HIP events on the
 * handle's stream): the SSV kernel alone, and the whole enqueue (model
 * padding copy + SSV + hit ordering). */
int havac_dev_last_run_ms(havac_dev *dev, float *ssv_kernel_ms, float *total_ms);

const char *havac_dev_last_error(havac_dev *dev);

/* ------------------------------------------------------------------------
 * Level 2: kernel ABI on caller-owned device memory (replaces HavacKernel)
 * ---------------------------------------------------------------------- */
typedef struct havac_ssv_ctx havac_ssv_ctx;

/* A context belongs to the HIP device that is current when it is created.  It
 * caches the scratch the launch needs (the expanded model and the sort's
 * temporary storage), so steady-state launches allocate nothing. */
int havac_ssv_ctx_create(havac_ssv_ctx **out);
void havac_ssv_ctx_destroy(havac_ssv_ctx *ctx);

/* Enqueue one SSV pass on `hip_stream` (a hipStream_t passed as void*; NULL is
 * the default stream) and return without waiting.  All pointers are DEVICE
 * pointers.
 *
 *   d_sequence   2-bit packed sequence, 8-byte aligned, nsymbols % 12288 == 0
 *   d_phmm       int8 [nrows][4], 4-byte aligned
 *   shard_index, shard_count
 *                the sequence is cut into shard_count runs of whole 12288-column
 *                segments and only the hits in run shard_index are computed and
 *                reported (the diagonals that reach the run are swept from their
 *                start, i.e. a left halo of nrows-1 columns is recomputed; no
 *                data is exchanged).  The shards' ordered lists, concatenated in
 *                shard order, are the ordered list of the whole.  0,1 = everything
 *   d_hits       receives up to hit_capacity records
 *   d_abort_flag optional uint32 the kernel polls; nonzero stops the run
 *
 * Mirrors HavacKernel(sequenceSegmentMemory, sequenceLengthInSegments,
 * phmmVectorMemory, phmmLengthInVectors, hitReportMemory, hitReportCountMemory)
 * at device/HavacHls.cpp:20-43; the hit count is returned by havac_ssv_finish. */
int havac_ssv_enqueue(havac_ssv_ctx *ctx, const uint8_t *d_sequence, uint64_t nsymbols,
                      const int8_t *d_phmm, uint32_t nrows, uint32_t shard_index,
                      uint32_t shard_count, uint64_t *d_hits, uint64_t hit_capacity,
                      const uint32_t *d_abort_flag, void *hip_stream);

/* Optional (no counterpart in the reference: one device holds the whole database there): a rank of a sharded run need
 * not hold the whole database.  havac_ssv_shard_window tells which columns shard `shard_index` of `shard_count` reads --
 * its own, the left halo of nrows - 1 columns and a few thousand more for the tiling; whole segments, clipped to the
 * database (host-only arithmetic).  havac_ssv_set_sequence_window declares that the d_sequence (and separator bitmap)
 * handed to the next passes hold columns [first_column, first_column + ncolumns) only, byte 0 of the buffer = column
 * first_column; nsymbols stays the database's total.  A window that does not cover what the shard reads is refused by
 * havac_ssv_enqueue.  (0, 0) = the buffer holds everything (the default).  C4: 35 MB per rank instead of 250 MB. */
int havac_ssv_shard_window(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index, uint32_t shard_count,
                           uint64_t *first_column, uint64_t *end_column);
int havac_ssv_set_sequence_window(havac_ssv_ctx *ctx, uint64_t first_column, uint64_t ncolumns);

/* Separator bitmap for the next passes (see havac_dev_write_separator_mask); a DEVICE pointer the caller keeps
 * alive, 2-byte aligned, nsymbols/16 bytes; NULL (the default) = none. */
int havac_ssv_set_separator_mask(havac_ssv_ctx *ctx, const uint8_t *d_pair_bitmap);

/* Optional: the HIP stream havac_ssv_finish orders the records on (NULL, the default = the stream of the enqueue).
 * With passes in flight a caller keeps all SSV kernels back to back on ONE stream and gives every pass its own
 * ordering stream: the kernel of pass k+1 then never waits behind the ordering of pass k (bench.py, havac_amd/dist.py).
 * No counterpart in the reference (one run at a time, host/HavacHwClient.cpp:150-170). */
int havac_ssv_set_order_stream(havac_ssv_ctx *ctx, void *hip_stream);

/* Per-cell trace: the counterpart of the reference's HAVAC_PER_CELL_DATA_TESTING build (device/PublicDefines.h:11,
 * device/HavacHls.cpp:388-399: every cell processor records prevValue, matchScore, cellValue, symbol, passesThreshold;
 * test/byCellComparator/byCellComparator.cpp:47-96 compares them with softSsv's).  While a trace window is set, passes
 * run a second instantiation of the SAME kernel body that writes one record per cell of rows [row0, row0 + nrows) x
 * columns [col0, col0 + ncols) into d_cells (nrows * ncols records, row-major, cleared by the caller; `written` tells
 * which cells the kernel visited).  A cell below a crossing on the same diagonal and inside the same four-step window
 * has `pending` set and no score: the kernel puts crossed cells back to 0 at the window's hit test, not before.
 * The hit list of such a pass is the usual one.  d_cells = NULL switches the trace off (the default). */
typedef struct havac_cell_record {
    uint8_t prev;      /* score of the diagonal's previous cell */
    int8_t match;      /* match score added */
    uint8_t score;     /* score afterwards; 0 after a crossing (device/HavacHls.cpp:384) */
    uint8_t hit;       /* passesThreshold */
    uint8_t symbol;    /* 0..3; 0xff under a separator (boundary mode) */
    uint8_t pending;   /* see above */
    uint8_t zero;
    uint8_t written;   /* 1 */
} havac_cell_record;
int havac_ssv_set_cell_trace(havac_ssv_ctx *ctx, havac_cell_record *d_cells, uint32_t row0, uint64_t col0,
                             uint32_t nrows, uint32_t ncols);

/* Completes the enqueued pass: waits for it, puts the shard's records in
 * d_hits into device order (on the enqueue's stream, or on the
 * ordering stream set above) and returns the
 * number of hits found.  If that exceeds hit_capacity only hit_capacity
 * records were kept and the result is HAVAC_E_HIT_OVERFLOW. */
int havac_ssv_finish(havac_ssv_ctx *ctx, uint64_t *hit_count_out);

/* havac_ssv_finish in two halves, for a caller with several contexts (several GPUs): _begin waits for the pass's hit count
 * and ENQUEUES the ordering, _end waits for the ordering and returns what havac_ssv_finish returns.  Begin all contexts,
 * then end all: the orderings run side by side instead of one after the other with a host wait in between
 * (havac_dev_wait does exactly this over the GPUs of a handle).  havac_ssv_finish = _begin + _end. */
int havac_ssv_finish_begin(havac_ssv_ctx *ctx);
int havac_ssv_finish_end(havac_ssv_ctx *ctx, uint64_t *hit_count_out);

/* Experiment knobs of the next passes (tools/, tests/): how the launch hands out work and how the records are ordered.
 * Every value: -1 = the library's own rule (the default).
 *   rows_per_block  > 0: EVERY tile is cut into uniform row blocks of that many rows (a multiple of 1024);
 *                   0 = never cut tiles into row blocks
 *   tiles_per_item  G >= 1: every wave walks a group of G adjacent tiles (short models; standard kernel); G < -1: groups of -G
 *                   tiles, but a partition's last round of wave slots as single tiles.  Any value but -1 keeps the library from
 *                   choosing the resident-table kernel by itself (havac_ssv_set_kernel_variant)
 *   block_tails     0 = never, 1 = the default rule, 2 = whatever the height of an item
 *   ordering        0 = always the generic radix sort, 1 = the bucket ordering (the default)
 * No counterpart in the reference. */
int havac_ssv_set_tuning(havac_ssv_ctx *ctx, int rows_per_block, int tiles_per_item, int block_tails, int ordering);
/* More of the same: the library's own rule for tall tiles.  Every value: -1 = keep the default.
 *   parts_log2       the launch's tiles are dealt to 2^parts_log2 partitions of adjacent tiles (0 ... 3; 3 = one per XCD)
 *   split_rounds_x4  the tiles of the launch's last split_rounds_x4 / 4 rounds of wave slots are cut by rows (default 6)
 *   short_rows       finest row block (a multiple of 1024; default 4096); models below twice that are never cut
 *   guide            a row block takes 1 / guide of the rows that are left (2 ... 16; default 2) */
int havac_ssv_set_split_tuning(havac_ssv_ctx *ctx, int parts_log2, int split_rounds_x4, int short_rows, int guide);

/* Which SSV kernel the next passes run.  -1 (the default): the library decides -- models of up to 256 (padded) rows take
 * ssv_resident_kernel: a tile of such a model is one to eight 32-row chunks and, handed out tile by tile, mostly prologue, so
 * ALL the model's match-word tables are built once per workgroup and kept in LDS, every wave walks a run of adjacent tiles, and
 * the runs are handed out in tapering rounds (havac_launch_plan: walk_len / walk_base); with a separator mask its second
 * instantiation, ssv_resident_kernel_masked (round 5); everything else, and every pass with a cell trace or a forced work
 * distribution (havac_ssv_set_tuning), takes the standard kernel, ssv_diag_kernel.
 * 0: always the standard kernel; 1: the resident-table kernel wherever it is valid (A/B, tests; with tiles_per_item = G >= 1:
 * runs of G tiles throughout).  The records are the same either way.  The reference's array runs at the same efficiency whatever
 * the model's height (README.md:4; device/HavacHls.cpp:220-319): this is what keeps short models near that.
 * havac_ssv_last_kernel_variant: what the last enqueued pass ran (0 standard, 1 resident tables). */
int havac_ssv_set_kernel_variant(havac_ssv_ctx *ctx, int variant);
int havac_ssv_last_kernel_variant(havac_ssv_ctx *ctx, int *variant);

/* How a launch of this shape would hand out its tiles (the same planner havac_ssv_enqueue uses; no device is touched, so
 * tests and tools can look at a plan anywhere).  `wave_slots`: waves the device holds at once (0 = an MI355X: 256 CUs x 24);
 * `tuning`: up to nine values as for havac_dev_set_tuning (NULL / -1 = the library's own rule).
 *   partitions      nparts runs of adjacent tiles of about equal work: partition k = tiles [part_begin[k], part_begin[k+1])
 *                   of the launch's ntiles tiles; workgroup i serves partition i mod nparts
 *   a partition's items, in this order: groups of tiles_per_group adjacent tiles (a wave walks a group; short models), single
 *                   tiles (its last single_tiles whole tiles, and what does not fill a group), then the row blocks of its
 *                   last cut_tiles tiles (all of them if it has fewer), row-block-major
 *   row blocks      block b < ncuts is rows [row_cut[b], row_cut[b+1]); from block ncuts on, uniform_rows rows each;
 *                   nrow_blocks blocks per cut tile (1: no tile of the launch is cut)
 * No counterpart in the reference. */
typedef struct havac_launch_plan {
    uint32_t nrows_padded, tile_begin, ntiles;
    uint32_t nparts, part_begin[9];
    uint32_t tiles_per_group, single_tiles, cut_tiles;
    uint32_t nrow_blocks, ncuts, uniform_rows, row_cut[33];
    uint32_t workgroups;
    uint32_t resident_kernel;   /* 1: the launch runs ssv_resident_kernel (short models; havac_ssv_set_kernel_variant): `workgroups` x 4 waves
                                   in walk_rounds rounds of walk_slots waves (the last round: the rest); wave i of round r walks
                                   walk_len[r] adjacent tiles from tile walk_base[r] + i * walk_len[r] on, clipped to ntiles */
    uint32_t walk_slots, walk_rounds, walk_len[8], walk_base[8];
} havac_launch_plan;
int havac_ssv_plan(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index, uint32_t shard_count, uint32_t wave_slots,
                   const int32_t *tuning, uint32_t ntuning, havac_launch_plan *plan_out);
/* The waves the context's device holds at once (compute units x 24): what havac_ssv_enqueue hands to the planner, and
 * what havac_ssv_plan must be given to reproduce a launch's plan on a device that is not a whole MI355X. */
int havac_ssv_wave_slots(havac_ssv_ctx *ctx, uint32_t *wave_slots);
/* How the last finished pass was ordered: *path = 0 radix sort, 1 bucket ordering, 2 bucket ordering given up for the
 * radix sort (a bucket too big for an LDS sort); the number of buckets and the largest one. */
int havac_ssv_last_ordering(havac_ssv_ctx *ctx, int *path, uint32_t *nbuckets, uint32_t *largest_bucket);

/* Sort `count` packed records in place into device order (for callers that merge
 * lists from elsewhere; the shards of havac_ssv_enqueue need no sorting, see above). */
int havac_ssv_sort_hits(havac_ssv_ctx *ctx, uint64_t *d_hits, uint64_t count, void *hip_stream);

/* The check a sharded run makes on the list it has gathered before it reports it (bench.py: distributed.parity; no
 * counterpart in the reference, which has one device per object, host/Havac.hpp:51).  `d_records`: `count` packed records
 * on the current device, the ranks' lists laid end to end in rank order; rank r contributed rank_counts[r] of them (the
 * counts must add up to `count`: HAVAC_E_LENGTH otherwise) and owns columns [span_begin[r], span_end[r]) (host arrays of
 * nranks <= 1024 entries; nranks == 0: order only).  One grid-stride pass on `hip_stream`, waited for before return;
 * no list-sized temporary (C4: 4.46e9 records = 36 GB, beyond 32-bit indexing).  Reports how many records are not
 * strictly greater than their predecessor in the reference's emission order -- segment, row, column
 * (device/HavacHls.cpp:151-152,264), so a duplicate counts too -- how many lie outside their rank's columns, and the
 * index of the first of each kind (UINT64_MAX: none). */
typedef struct havac_order_report {
    uint64_t records;
    uint64_t out_of_order, first_out_of_order;
    uint64_t out_of_span, first_out_of_span;
} havac_order_report;
int havac_ssv_check_order(const uint64_t *d_records, uint64_t count, const uint64_t *rank_counts, const uint64_t *span_begin,
                          const uint64_t *span_end, uint32_t nranks, void *hip_stream, havac_order_report *report_out);

/* Without waiting: 1 if the enqueued pass -- kernel and ordering -- has completed on the device (havac_ssv_finish will not block),
 * 0 if not yet, negative on error. */
int havac_ssv_query(havac_ssv_ctx *ctx);
/* Waits until the enqueued pass has READ its inputs: the model (its padded copy is made by the pass's first kernel) and, with
 * sequence_too != 0, the sequence (read by the SSV kernel).  The caller may overwrite those buffers afterwards, while the rest
 * of the pass still runs.  Returns at once when nothing is enqueued. */
int havac_ssv_wait_inputs(havac_ssv_ctx *ctx, int sequence_too);

/* After the stream has been synchronised: device time of the last enqueue's
 * SSV kernel and of the whole enqueue, from HIP events recorded on that stream. */
int havac_ssv_last_ms(havac_ssv_ctx *ctx, float *ssv_kernel_ms, float *total_ms);

/* Number of DP cells of the given shard (its columns x nrows; the recomputed
 * halo is not counted), for GCUPS accounting. */
uint64_t havac_ssv_shard_cells(uint64_t nsymbols, uint32_t nrows, uint32_t shard_index,
                               uint32_t shard_count);

/* The half-open range of columns shard `shard_index` owns (multiples of 12288).
 * Host-only arithmetic (no device needed). */
int havac_ssv_shard_columns(uint64_t nsymbols, uint32_t shard_index, uint32_t shard_count,
                            uint64_t *col_begin, uint64_t *col_end);

const char *havac_ssv_ctx_last_error(havac_ssv_ctx *ctx);

/* ------------------------------------------------------------------------
 * Level 2b: passes in flight
 * ---------------------------------------------------------------------- */
/* No counterpart in the reference, whose API is one run at a time (host/HavacHwClient.cpp:141-157: invoke, wait, list).  A pipe
 * keeps up to `depth` passes in flight on the device that is current when it is created, each in a slot of its own -- a context
 * and a hit buffer of hit_capacity records -- and each, from its first kernel to the ordering of its records, on ONE of two
 * high-priority streams that consecutive passes alternate between (kernel_streams: -1 = that; 1 = every pass on the same stream;
 * 2 .. 4 = that many streams taken in turn, three and four for experiments): while the host waits for pass k and its records are
 * ordered and gathered, the kernel of pass k + 1 runs -- and has started while kernel k was draining.  depth 1 is the reference's
 * behaviour.
 *
 *   havac_pipe_submit   enqueues one whole pass (as havac_ssv_enqueue; all pointers are DEVICE pointers the caller keeps alive)
 *                       behind what `caller_stream` holds now (the inputs' producer; HAVAC_NO_STREAM: nothing to wait for);
 *                       HAVAC_E_LOGIC when every slot is in flight
 *   havac_pipe_collect  completes the OLDEST pass in flight: waits for it and returns the hits it found and its records in device
 *                       order -- the slot's hit buffer, or, with a gather set, on rank 0 the whole list of all ranks in one
 *                       buffer (NULL elsewhere).  The records stay valid until that slot is submitted again (depth passes later);
 *                       `caller_stream` has been made to wait for them.  With a gather set a rank whose pass failed still takes
 *                       part in the count exchange, and every rank then returns an error: none is left inside a collective.
 *   havac_pipe_poll     1 if havac_pipe_collect would not block for the device, 0 if it would, negative on error
 *   havac_pipe_wait_inputs  waits until every pass in flight has read the model (and, sequence_too != 0, the sequence): the caller
 *                       may then overwrite those buffers for the next submit
 *   havac_pipe_context  a slot's context (for the tuning and window setters of level 2); which = -1: of the last collected pass
 *   havac_pipe_set_gather   the communicator of a sharded run (level 3): collect() then gathers every pass's records to rank 0
 *   havac_pipe_wait_gathers, havac_pipe_gather_times   the records of the last collected pass may still be travelling: waits for
 *                       them with the gather's deadline; device milliseconds of the gathers so far (and forgets them)
 *   havac_pipe_release  gives every buffer back but the one that holds the records the last collect returned (valid until
 *                       havac_pipe_destroy); nothing may be in flight; the pipe cannot be submitted to again */
typedef struct havac_pipe havac_pipe;
typedef struct havac_gather havac_gather;
#define HAVAC_NO_STREAM ((void *)(intptr_t)-1)      /* as `caller_stream`: the inputs are in place, there is nothing to wait for */
int havac_pipe_create(uint32_t depth, uint64_t hit_capacity, int kernel_streams, havac_pipe **out);
void havac_pipe_destroy(havac_pipe *pipe);
int havac_pipe_submit(havac_pipe *pipe, const uint8_t *d_sequence, uint64_t nsymbols, const int8_t *d_phmm, uint32_t nrows,
                      uint32_t shard_index, uint32_t shard_count, const uint32_t *d_abort_flag, void *caller_stream);
int havac_pipe_collect(havac_pipe *pipe, uint64_t *found_out, const uint64_t **d_records_out, uint64_t *nrecords_out, void *caller_stream);
/* `nsteps` passes of the same inputs, as many in flight as the pipe is deep, all complete on return (submit, and collect the
 * oldest once every slot is in flight); kernel_ms / total_ms: nsteps entries each or NULL; the last pass's result as
 * havac_pipe_collect returns it.  A timed region without the caller's language in it. */
int havac_pipe_run(havac_pipe *pipe, uint32_t nsteps, const uint8_t *d_sequence, uint64_t nsymbols, const int8_t *d_phmm, uint32_t nrows,
                   uint32_t shard_index, uint32_t shard_count, void *caller_stream, float *kernel_ms, float *total_ms,
                   uint64_t *found_out, const uint64_t **d_records_out, uint64_t *nrecords_out);
int havac_pipe_poll(havac_pipe *pipe);
int havac_pipe_wait_inputs(havac_pipe *pipe, int sequence_too);
uint32_t havac_pipe_depth(havac_pipe *pipe);
uint32_t havac_pipe_in_flight(havac_pipe *pipe);
int havac_pipe_used_two_streams(havac_pipe *pipe);
int havac_pipe_streams_used(havac_pipe *pipe);      /* how many streams the last submit took its turn over */
havac_ssv_ctx *havac_pipe_context(havac_pipe *pipe, int which);
int havac_pipe_set_gather(havac_pipe *pipe, havac_gather *gather);
int havac_pipe_wait_gathers(havac_pipe *pipe);
int havac_pipe_gather_times(havac_pipe *pipe, float *out_ms, uint32_t capacity, uint32_t *count);
int havac_pipe_last_ms(havac_pipe *pipe, float *ssv_kernel_ms, float *total_ms);
int havac_pipe_release(havac_pipe *pipe);
const char *havac_pipe_last_error(havac_pipe *pipe);

/* ------------------------------------------------------------------------
 * Level 3: the one exchange of a sharded run -- an RCCL gather of the hit records to rank 0
 * ---------------------------------------------------------------------- */
/* BASELINE.json north_star: "... a thin C-ABI HIP layer ... shards embarrassingly across the 8 GPUs of one node with only
 * an RCCL gather of HavacHit records over xGMI at the end".  No counterpart in the reference (one deviceIndex per object,
 * host/Havac.hpp:51).  One process per GPU: every rank runs havac_ssv_enqueue / havac_ssv_finish on its shard; because
 * shards are whole segments, the ranks' ordered lists laid end to end in rank order are the reference's device order
 * (device/HavacHls.cpp:151-152,264), so rank 0 receives and never sorts.  librccl is bound at run time (the copy already in
 * the process, else the system's): a single-GPU caller never loads it.
 *
 *   havac_gather_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks over any side
 *                            channel (a file, a socket, torch.distributed's store)
 *   havac_gather_create      collective: every rank, on its own current device, with the same id (ncclCommInitRank)
 *   havac_gather_counts      collective, step 1 of a gather: ncclAllGather of one int64 per rank -- the records its pass left, or
 *                            a negative number if its pass failed; counts_out (host, `world` entries) is the same on every rank.
 *                            The one host wait of a gather.  A rank that sees a negative count skips step 2 (every rank does).
 *   havac_gather_records     collective, step 2: rank r > 0 sends exactly its count records (ncclSend), rank 0 receives each list
 *                            straight into d_out at the exclusive-scan offset of its rank (ncclRecv, one group) and copies its own:
 *                            d_out[0 .. sum(counts)) is the whole list in device order.  Nothing is padded, nothing concatenated
 *                            afterwards; rank 0 sizes d_out from counts_out.  Enqueued behind what `hip_stream` holds, on the
 *                            communicator's own low-priority stream; `hip_stream` is made to wait for it, the host is not.
 *                            d_out / out_capacity are ignored on ranks > 0.  A receive buffer that is too small is refused on
 *                            rank 0 before anything is posted (HAVAC_E_LENGTH) and leaves the communicator unusable.
 *   havac_gather_destroy     collective in the good case (ncclCommDestroy); aborts a broken communicator.
 *   havac_gather_set_deadline  every host wait of this communicator -- for the counts, in havac_gather_wait, in
 *                            havac_gather_destroy -- gives up after timeout_ms (0, the default: never) with HAVAC_E_TIMEOUT and a
 *                            message that names this rank and the stage: a rank that died, or never reached the collective, then
 *                            costs the others `timeout_ms`, not the job's whole time limit.  The communicator is unusable afterwards.
 *   havac_gather_wait        waits (with the deadline) for what the last havac_gather_records enqueued; a caller that lets the
 *                            records travel behind its next pass calls this before it reuses the buffers or ends its timed region.
 *   havac_gather_use_library names the collective library to bind INSTEAD of librccl.so.1, before the first havac_gather_* call
 *                            of the process binds one (afterwards: HAVAC_E_LOGIC unless it is the same path; NULL = RCCL).  For
 *                            rehearsals of several ranks on ONE GPU, which RCCL refuses: tests/native/rccl_standin.cpp implements
 *                            the eleven bound entry points over shared memory.  An argument, not an environment variable. */
#define HAVAC_GATHER_ID_BYTES 128
int havac_gather_rccl_version(int *version);
int havac_gather_unique_id(uint8_t id[HAVAC_GATHER_ID_BYTES]);
int havac_gather_create(uint32_t rank, uint32_t world, const uint8_t id[HAVAC_GATHER_ID_BYTES], havac_gather **out);
int havac_gather_counts(havac_gather *g, int64_t my_count, int64_t *counts_out, void *hip_stream);
int havac_gather_records(havac_gather *g, const uint64_t *d_records, uint64_t *d_out, uint64_t out_capacity, void *hip_stream);
int havac_gather_set_deadline(havac_gather *g, uint32_t timeout_ms);
int havac_gather_wait(havac_gather *g);
int havac_gather_use_library(const char *path);
int havac_gather_info(havac_gather *g, uint32_t *rank, uint32_t *world);
const char *havac_gather_last_error(havac_gather *g);
void havac_gather_destroy(havac_gather *g);

/* Library self-description, e.g. "havac_dev 0.1 gfx950". */
const char *havac_dev_version(void);

#ifdef __cplusplus
}
#endif
#endif
