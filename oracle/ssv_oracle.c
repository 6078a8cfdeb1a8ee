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

#ifndef HAVAC_BUILD_FLAGS
#define HAVAC_BUILD_FLAGS "?"
#endif
/* compiler and flags of this object (bench.py: cpu_baseline.compiler) */
const char *havac_oracle_build_info(void) { return "gcc " __VERSION__ " " HAVAC_BUILD_FLAGS; }

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

/* ---- AVX2 diagonal bands ------------------------------------------------ */
#if defined(__x86_64__)
#include <immintrin.h>

/* A cell depends only on its own diagonal d = s - p, so the matrix is cut into bands of
 * FAST_BAND consecutive diagonals that never exchange anything: a band keeps one score byte per diagonal (L1
 * resident), and row p updates it in place from the symbols at columns d + p.  No halo, exactly n * nrows cell
 * updates.  Bands are dealt to the threads round robin; every thread collects its records in a growing buffer, sorts
 * them by device-order key, and the lists are merged at the end. */
#define FAST_BAND 16384

struct fast_job {
    const uint8_t *symbols; uint64_t n; const int8_t *model; uint64_t nrows;
    uint64_t nbands; int thread, nthreads;
    uint64_t *keys; uint64_t count, room; int failed;
};

static uint64_t device_key_of(uint64_t row, uint64_t column) {
    return (column / HAVAC_ORACLE_SEGMENT) << 38 | row << 14 | (column % HAVAC_ORACLE_SEGMENT);
}

static void fast_emit(struct fast_job *j, uint64_t row, uint64_t column) {
    if (j->count == j->room) {
        uint64_t room = j->room ? 2 * j->room : 4096;
        uint64_t *grown = (uint64_t *)realloc(j->keys, (size_t)room * sizeof(uint64_t));
        if (!grown) { j->failed = 1; return; }
        j->keys = grown; j->room = room;
    }
    j->keys[j->count++] = device_key_of(row, column);
}

/* diagonals [d0, d0 + FAST_BAND) of the matrix, d0 may be negative */
__attribute__((target("avx2")))
static void fast_band(struct fast_job *j, int64_t d0, uint8_t *score) {
    const int64_t n = (int64_t)j->n, d1 = d0 + FAST_BAND;
    memset(score, 0, FAST_BAND + 32);
    const __m256i zero = _mm256_setzero_si256();
    const __m256i limit = _mm256_set1_epi16(255);
    for (int64_t p = 0; p < (int64_t)j->nrows; p++) {
        int64_t lo = d0 > -p ? d0 : -p;                  /* column d + p >= 0 */
        int64_t hi = d1 < n - p ? d1 : n - p;            /* column d + p < n */
        if (lo >= hi) { if (d0 + p >= n) break; continue; }
        const int8_t *scores = j->model + 4 * p;
        int32_t row_word;
        memcpy(&row_word, scores, 4);
        const __m128i lut = _mm_set1_epi32(row_word);    /* bytes A,C,G,T repeated: pshufb index 0..3 */
        const uint8_t *sym = j->symbols + p;             /* sym[d] is the symbol of column d + p */
        int64_t d = lo;
        for (; d + 16 <= hi; d += 16) {
            uint8_t *cell = score + (d - d0);
            const __m128i sy = _mm_loadu_si128((const __m128i *)(sym + d));
            const __m256i m = _mm256_cvtepi8_epi16(_mm_shuffle_epi8(lut, sy));
            const __m256i old = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)cell));
            __m256i t = _mm256_add_epi16(old, m);
            const __m256i hit = _mm256_cmpgt_epi16(t, limit);       /* t >= 256 */
            t = _mm256_andnot_si256(hit, _mm256_max_epi16(t, zero));
            const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi16(t, t), 0x08);
            _mm_storeu_si128((__m128i *)cell, _mm256_castsi256_si128(packed));
            unsigned mask = (unsigned)_mm256_movemask_epi8(hit);
            while (mask) {
                const unsigned k = (unsigned)__builtin_ctz(mask) / 2;
                mask &= ~(3u << (2 * k));
                fast_emit(j, (uint64_t)p, (uint64_t)(d + k + p));
            }
        }
        for (; d < hi; d++) {                            /* ragged end of an edge band */
            int t = (int)score[d - d0] + (int)scores[sym[d]];
            uint8_t v = (uint8_t)t;
            if (t < 0) v = 0;
            if (t >= 256) { v = 0; fast_emit(j, (uint64_t)p, (uint64_t)(d + p)); }
            score[d - d0] = v;
        }
    }
}

static void *fast_main(void *arg) {
    struct fast_job *j = (struct fast_job *)arg;
    uint8_t *score = (uint8_t *)malloc(FAST_BAND + 64);
    if (!score) { j->failed = 1; return NULL; }
    const int64_t first = -((int64_t)j->nrows - 1);
    for (uint64_t b = (uint64_t)j->thread; b < j->nbands && !j->failed; b += (uint64_t)j->nthreads)
        fast_band(j, first + (int64_t)(b * FAST_BAND), score);
    free(score);
    if (!j->failed && j->count) qsort(j->keys, (size_t)j->count, sizeof(uint64_t), cmp_u64);
    return NULL;
}

int64_t havac_oracle_ssv_fast(const uint8_t *symbols, uint64_t n, const int8_t *model,
                              uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads) {
    if (!__builtin_cpu_supports("avx2")) return havac_oracle_ssv_mt(symbols, n, model, nrows, hits, cap, nthreads);
    if (n == 0 || nrows == 0) return 0;
    if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads < 1) nthreads = 1;
    const uint64_t nbands = (n + nrows - 1 + FAST_BAND - 1) / FAST_BAND;
    if ((uint64_t)nthreads > nbands) nthreads = (int)nbands;
    /* the 16-byte symbol loads of the last columns read past the end: work on a padded copy */
    uint8_t *padded = (uint8_t *)malloc((size_t)n + 64);
    struct fast_job *jobs = (struct fast_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof *tids);
    if (!padded || !jobs || !tids) { free(padded); free(jobs); free(tids); return -1; }
    memcpy(padded, symbols, (size_t)n);
    memset(padded + n, 0, 64);
    for (int t = 0; t < nthreads; t++)
        jobs[t] = (struct fast_job){padded, n, model, nrows, nbands, t, nthreads, NULL, 0, 0, 0};
    for (int t = 0; t < nthreads; t++) pthread_create(&tids[t], NULL, fast_main, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(tids[t], NULL);
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        if (jobs[t].failed) total = -1;
        else if (total >= 0) total += (int64_t)jobs[t].count;
    }
    if (total >= 0) {
        /* merge the sorted lists (keys are unique: one per cell); only the first `cap` records are stored */
        uint64_t *at = (uint64_t *)calloc((size_t)nthreads, sizeof *at);
        if (!at) total = -1;
        for (uint64_t out = 0; total >= 0 && out < (uint64_t)total && out < cap; out++) {
            int best = -1;
            for (int t = 0; t < nthreads; t++)
                if (at[t] < jobs[t].count && (best < 0 || jobs[t].keys[at[t]] < jobs[best].keys[at[best]])) best = t;
            const uint64_t key = jobs[best].keys[at[best]++];
            const uint64_t column = (key >> 38) * HAVAC_ORACLE_SEGMENT + (key & 0x3fffull);
            hits[out] = havac_oracle_pack_hit((uint32_t)((key >> 14) & 0xffffffu), column);
        }
        free(at);
    }
    for (int t = 0; t < nthreads; t++) free(jobs[t].keys);
    free(padded); free(jobs); free(tids);
    return total;
}
#else
int64_t havac_oracle_ssv_fast(const uint8_t *symbols, uint64_t n, const int8_t *model,
                              uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads) {
    return havac_oracle_ssv_mt(symbols, n, model, nrows, hits, cap, nthreads);
}
#endif

/* ---- per-cell records -------------------------------------------------- */

/* What test/softSsv/SoftSsv.cpp:59-65 records per cell in a HAVAC_PER_CELL_DATA_TESTING build (prevValue, matchScore,
 * cellValue, symbol, passesThreshold), for the cells of rows [row0, row0 + h) x columns [col0, col0 + w), in the 8-byte
 * layout of include/havac_dev.h (havac_cell_record; `pending` is always 0 here).  out: h * w records, row-major.
 * The whole matrix above the window is swept (rows outer, columns inner, as in sweep()). */
int havac_oracle_cells(const uint8_t *symbols, uint64_t n, const int8_t *model, uint64_t nrows,
                       uint64_t row0, uint64_t col0, uint64_t h, uint64_t w, uint8_t *out) {
    if (row0 + h > nrows || col0 + w > n) return -2;
    uint8_t *row = (uint8_t *)calloc((size_t)n, 1);
    if (!row) return -1;
    for (uint64_t p = 0; p < row0 + h; p++) {
        const int8_t *scores = model + 4 * p;
        uint8_t upper_left = 0;
        for (uint64_t s = 0; s < n; s++) {
            uint8_t above = row[s];
            int hit = 0;
            int8_t match = scores[symbols[s]];
            uint8_t v = havac_oracle_cell(upper_left, match, &hit);
            if (p >= row0 && s >= col0 && s < col0 + w) {
                uint8_t *rec = out + 8 * ((p - row0) * w + (s - col0));
                rec[0] = upper_left; rec[1] = (uint8_t)match; rec[2] = v; rec[3] = (uint8_t)hit;
                rec[4] = symbols[s]; rec[5] = 0; rec[6] = 0; rec[7] = 1;
            }
            upper_left = above;
            row[s] = v;
        }
    }
    free(row);
    return 0;
}
