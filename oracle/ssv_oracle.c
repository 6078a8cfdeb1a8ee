/*
 * ssv_oracle.c -- CPU restatement of HAVAC's SSV hot path (see ssv_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and the
 * cpu_baseline leg of bench.py.  Never linked into the product library.
 *
 * Written from the reference's behaviour, not its text: the sweep below walks
 * columns left to right and carries the upper-left neighbour in a register,
 * where test/softSsv/SoftSsv.cpp:31-62 walks right to left over one row
 * buffer.  Both visit every (row, column) once with the previous row's score
 * of the column to the left, so the cell values and the set of hits are the
 * same; `make -C oracle check` and tests/test_oracle.py prove it against the
 * reference object built in oracle/_ref.
 */
#include "ssv_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ---- single cell ------------------------------------------------------- */

/* test/softSsv/SoftSsv.cpp:36-47 : int32 sum of the int8 match score and the
 * uint8 previous score; negative clamps to 0; 256 or more resets to 0 and is
 * a hit; anything else is kept. */
uint8_t havac_oracle_cell(uint8_t prev, int8_t match, int *hit) {
    int t = (int)prev + (int)match;
    *hit = 0;
    if (t < 0) return 0;
    if (t >= 256) { *hit = 1; return 0; }
    return (uint8_t)t;
}

/* device/HavacHls.cpp:376-386 : 9-bit add of two 8-bit patterns; the match
 * score's sign bit against the carry-out decides reset, carry without sign is
 * the threshold crossing. */
uint8_t havac_oracle_cell_carry(uint8_t prev, uint8_t match_bits, int *hit) {
    unsigned sum9 = (unsigned)prev + (unsigned)match_bits;
    unsigned sign = (match_bits >> 7) & 1u;
    unsigned carry = (sum9 >> 8) & 1u;
    *hit = (int)(carry & (sign ^ 1u));
    return (carry != sign) ? 0 : (uint8_t)(sum9 & 0xffu);
}

/* ---- hit record -------------------------------------------------------- */

/* device/HitReporting.cpp:421-430 (bits(14,0), bits(40,14), bits(64,40)) */
uint64_t havac_oracle_pack_hit(uint32_t row, uint64_t column) {
    uint64_t seg = column / HAVAC_ORACLE_SEGMENT;
    uint64_t in_seg = column % HAVAC_ORACLE_SEGMENT;
    return (in_seg & 0x3fffull) | ((seg & 0x3ffffffull) << 14) | ((uint64_t)(row & 0xffffffu) << 40);
}

/* host/Havac.cpp:155-163 */
void havac_oracle_unpack_hit(uint64_t record, uint32_t *row, uint64_t *column) {
    uint64_t in_seg = record & 0x3fffull;
    uint64_t seg = (record >> 14) & 0x3ffffffull;
    *column = seg * HAVAC_ORACLE_SEGMENT + in_seg;
    *row = (uint32_t)(record >> 40);
}

/* ---- 2-bit packing ----------------------------------------------------- */

/* host/sequence/SequencePreprocessor.cpp:46-57 */
void havac_oracle_unpack_2bit(const uint8_t *packed, uint64_t nsymbols, uint8_t *symbols) {
    for (uint64_t i = 0; i < nsymbols; i++)
        symbols[i] = (uint8_t)((packed[i >> 2] >> ((i & 3u) * 2u)) & 3u);
}

void havac_oracle_pack_2bit(const uint8_t *symbols, uint64_t nsymbols, uint8_t *packed) {
    memset(packed, 0, (size_t)((nsymbols + 3) / 4));
    for (uint64_t i = 0; i < nsymbols; i++)
        packed[i >> 2] |= (uint8_t)((symbols[i] & 3u) << ((i & 3u) * 2u));
}

/* ---- the sweep --------------------------------------------------------- */

/* Rows outer, columns inner over symbols[0..n).  Column 0 of this span sees a
 * previous score of 0 on every row (for the whole matrix that is the rule of
 * SoftSsv.cpp:38; for a window it is harmless because every kept cell's
 * diagonal starts on row 0 inside the span).  Hits whose span column is below
 * `keep_from` are dropped; the rest are stored with `col_base` added. */
static int64_t sweep(const uint8_t *symbols, uint64_t n, const int8_t *model, uint64_t nrows,
                     uint64_t keep_from, uint64_t col_base, uint64_t *hits, uint64_t cap) {
    if (n == 0 || nrows == 0) return 0;
    uint8_t *row = (uint8_t *)calloc((size_t)n, 1); /* row -1 is all zero: SoftSsv.cpp:23 */
    if (!row) return -1;
    uint64_t found = 0;
    for (uint64_t p = 0; p < nrows; p++) {
        const int8_t *scores = model + 4 * p;
        uint8_t upper_left = 0; /* column 0: SoftSsv.cpp:38 */
        for (uint64_t s = 0; s < n; s++) {
            uint8_t above = row[s];
            int t = (int)upper_left + (int)scores[symbols[s]];
            upper_left = above;
            uint8_t v = (uint8_t)t;
            if (t < 0) v = 0;
            if (t >= 256) {
                v = 0;
                if (s >= keep_from) {
                    if (found < cap) hits[found] = havac_oracle_pack_hit((uint32_t)p, col_base + s);
                    found++;
                }
            }
            row[s] = v;
        }
    }
    free(row);
    return (int64_t)found;
}

int64_t havac_oracle_ssv(const uint8_t *symbols, uint64_t n, const int8_t *model,
                         uint64_t nrows, uint64_t *hits, uint64_t cap) {
    return sweep(symbols, n, model, nrows, 0, 0, hits, cap);
}

int64_t havac_oracle_ssv_window(const uint8_t *symbols, uint64_t n, const int8_t *model,
                                uint64_t nrows, uint64_t col_begin, uint64_t col_end,
                                uint64_t *hits, uint64_t cap) {
    if (col_end > n) col_end = n;
    if (col_begin >= col_end || nrows == 0) return 0;
    uint64_t halo = nrows - 1;
    uint64_t start = col_begin > halo ? col_begin - halo : 0;
    return sweep(symbols + start, col_end - start, model, nrows, col_begin - start, start, hits, cap);
}

/* ---- ordering ---------------------------------------------------------- */

static uint64_t device_key(uint64_t rec) {
    /* (segment, row, column-in-segment) as one integer */
    return ((rec >> 14) & 0x3ffffffull) << 38 | (rec >> 40) << 14 | (rec & 0x3fffull);
}
static int cmp_device(const void *a, const void *b) {
    uint64_t x = device_key(*(const uint64_t *)a), y = device_key(*(const uint64_t *)b);
    return (x > y) - (x < y);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

void havac_oracle_sort_device_order(uint64_t *hits, uint64_t count) {
    qsort(hits, (size_t)count, sizeof(uint64_t), cmp_device);
}

/* the packed layout already has row above segment above column */
void havac_oracle_sort_row_major(uint64_t *hits, uint64_t count) {
    qsort(hits, (size_t)count, sizeof(uint64_t), cmp_u64);
}

/* ---- threads ----------------------------------------------------------- */

struct block_job {
    const uint8_t *symbols; uint64_t n; const int8_t *model; uint64_t nrows;
    uint64_t begin, end; uint64_t *hits; uint64_t cap; int64_t found;
};

static void *block_main(void *arg) {
    struct block_job *j = (struct block_job *)arg;
    j->found = havac_oracle_ssv_window(j->symbols, j->n, j->model, j->nrows, j->begin, j->end,
                                       j->hits, j->cap);
    return NULL;
}

int64_t havac_oracle_ssv_mt(const uint8_t *symbols, uint64_t n, const int8_t *model,
                            uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads) {
    if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > n) nthreads = n ? (int)n : 1;
    struct block_job *jobs = (struct block_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof *tids);
    if (!jobs || !tids) { free(jobs); free(tids); return -1; }
    uint64_t per = (n + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        uint64_t b = per * (uint64_t)t, e = b + per;
        if (b > n) b = n;
        if (e > n) e = n;
        jobs[t] = (struct block_job){symbols, n, model, nrows, b, e, NULL, cap, 0};
        jobs[t].hits = (uint64_t *)malloc((size_t)(cap ? cap : 1) * sizeof(uint64_t));
        if (!jobs[t].hits) total = -1;
    }
    if (total == 0) {
        for (int t = 0; t < nthreads; t++) pthread_create(&tids[t], NULL, block_main, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(tids[t], NULL);
        uint64_t stored = 0;
        for (int t = 0; t < nthreads; t++) {
            if (jobs[t].found < 0) { total = -1; break; }
            uint64_t have = (uint64_t)jobs[t].found < cap ? (uint64_t)jobs[t].found : cap;
            for (uint64_t i = 0; i < have && stored < cap; i++) hits[stored++] = jobs[t].hits[i];
            total += jobs[t].found;
        }
        if (total >= 0) havac_oracle_sort_device_order(hits, stored);
    }
    for (int t = 0; t < nthreads; t++) free(jobs[t].hits);
    free(jobs); free(tids);
    return total;
}
