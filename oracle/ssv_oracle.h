/*
 * ssv_oracle.h -- CPU restatement of HAVAC's SSV hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The shipped path
 * (havac_amd/csrc, libhavac_dev.so) never links or calls it.
 *
 * Parity pinning: every function here is checked in this container against the
 * reference's own test/softSsv/SoftSsv.cpp compiled unmodified from where it
 * lies under /root/reference (oracle/Makefile target `_ref`), and against the
 * committed fixtures under tests/golden/ that the same build produced
 * (tests/golden/make_golden.py).  The reference ships no golden vectors of its
 * own.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef HAVAC_SSV_ORACLE_H
#define HAVAC_SSV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Columns per hardware sequence segment: NUM_CELL_GROUPS * CELLS_PER_GROUP,
 * device/PublicDefines.h:18-22.  Baked into the packed hit format. */
#define HAVAC_ORACLE_SEGMENT 12288u

/* One SSV cell, arithmetic form.  test/softSsv/SoftSsv.cpp:36-47.
 * Returns the new cell score, sets *hit to 1 when the sum reached 256. */
uint8_t havac_oracle_cell(uint8_t prev, int8_t match, int *hit);

/* One SSV cell, the FPGA's 9-bit carry/sign form.  device/HavacHls.cpp:370-386.
 * Must agree with havac_oracle_cell on all 65536 inputs (tests check this). */
uint8_t havac_oracle_cell_carry(uint8_t prev, uint8_t match_bits, int *hit);

/* Packed 64-bit hit record: [13:0] column inside its 12288-wide segment,
 * [39:14] segment index, [63:40] global model row.
 * device/HitReporting.cpp:421-430; decoder host/Havac.cpp:155-163. */
uint64_t havac_oracle_pack_hit(uint32_t row, uint64_t column);
void havac_oracle_unpack_hit(uint64_t record, uint32_t *row, uint64_t *column);

/* 2-bit packed sequence -> one symbol per byte.  Symbol i sits in byte i/4 at
 * bit (i%4)*2.  host/sequence/SequencePreprocessor.cpp:46-57,
 * device/HavacHls.cpp:421-425. */
void havac_oracle_unpack_2bit(const uint8_t *packed, uint64_t nsymbols, uint8_t *symbols);
void havac_oracle_pack_2bit(const uint8_t *symbols, uint64_t nsymbols, uint8_t *packed);

/* Whole-matrix SSV.  Same semantics as softSsvThreshold256
 * (test/softSsv/SoftSsv.cpp:15-67): rows are the concatenated model, columns
 * the concatenated padded sequence, row 0 and column 0 see a previous score
 * of 0, threshold 256 resets the cell and reports it.
 *   symbols : n bytes, values 0..3
 *   model   : nrows*4 int8, [row][A,C,G,T]
 *   hits    : receives packed records (havac_oracle_pack_hit) in row-major,
 *             column-ascending order; only the first `cap` are stored
 * Returns the number of hits found (may exceed cap), or -1 if memory ran out
 * (the reference's errorCode = -1, SoftSsv.cpp:24-29). */
int64_t havac_oracle_ssv(const uint8_t *symbols, uint64_t n, const int8_t *model,
                         uint64_t nrows, uint64_t *hits, uint64_t cap);

/* The same answer for the column range [col_begin, col_end) only, computed
 * from a left halo of nrows-1 columns (a cell depends
 * only on its own diagonal).  Used for window checks at sizes the full oracle
 * cannot finish, and as the shard model for the multi-GPU tests. */
int64_t havac_oracle_ssv_window(const uint8_t *symbols, uint64_t n, const int8_t *model,
                                uint64_t nrows, uint64_t col_begin, uint64_t col_end,
                                uint64_t *hits, uint64_t cap);

/* Multi-threaded whole-matrix SSV: column blocks with the halo above, one
 * pthread per block, results concatenated then put in device order.
 * nthreads <= 0 picks the number of online CPUs. */
int64_t havac_oracle_ssv_mt(const uint8_t *symbols, uint64_t n, const int8_t *model,
                            uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads);

/* The same answer as havac_oracle_ssv_mt, computed with AVX2 (16 cells per instruction) on bands of 16384
 * diagonals whose scores stay in the L1 cache (a cell depends only on its own diagonal: no halo), bands dealt to
 * pthreads.  Falls back to havac_oracle_ssv_mt on a CPU without AVX2.
 * A different route to the same numbers: tests/test_oracle.py checks it against havac_oracle_ssv and the
 * reference object; the GPU tests use it to compare WHOLE hit lists at BASELINE.json's full sizes. */
int64_t havac_oracle_ssv_fast(const uint8_t *symbols, uint64_t n, const int8_t *model,
                              uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads);

/* Sort packed records into the FPGA's emission order: segment ascending, then
 * row ascending, then column-in-segment ascending.
 * device/HavacHls.cpp:151-152,264; device/HitReporting.cpp:178-337. */
void havac_oracle_sort_device_order(uint64_t *hits, uint64_t count);

/* "gcc <version> <flags>": how this library was built. */
const char *havac_oracle_build_info(void);

/* Sort packed records by (row, column): the order the test comparisons use. */
void havac_oracle_sort_row_major(uint64_t *hits, uint64_t count);

/* Per-cell records of rows [row0, row0 + h) x columns [col0, col0 + w): what test/softSsv/SoftSsv.cpp:59-65 records in a
 * HAVAC_PER_CELL_DATA_TESTING build, in the 8-byte layout of include/havac_dev.h (havac_cell_record).  out: h * w * 8
 * bytes, row-major.  0, or negative on a bad window / failed allocation. */
int havac_oracle_cells(const uint8_t *symbols, uint64_t n, const int8_t *model, uint64_t nrows,
                       uint64_t row0, uint64_t col0, uint64_t h, uint64_t w, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
