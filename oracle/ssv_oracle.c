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

/* ---- AVX2 tiles --------------------------------------------------------- */
#if defined(__x86_64__)
#include <immintrin.h>

/* tile width in columns: the two row buffers of a tile (tile + nrows bytes each) stay in L1/L2, and the left halo
 * of nrows-1 columns every tile recomputes stays a modest share of it */
static uint64_t fast_tile_width(uint64_t nrows) {
    uint64_t w = 8192;
    while (w < 2 * nrows) w *= 2;
    return w;
}

struct fast_job {
    const uint8_t *symbols; uint64_t n; const int8_t *model; uint64_t nrows;
    uint64_t tile, tile_begin, tile_end;      /* tiles [tile_begin, tile_end) of `tile` columns */
    uint64_t *hits; uint64_t cap; int64_t found;
};

/* one tile: columns [a, b) of the matrix, swept from column a - (nrows-1) (or 0); two row buffers in L1 */
__attribute__((target("avx2")))
static int64_t fast_tile(const struct fast_job *j, uint64_t a, uint64_t b, uint8_t *buf0, uint8_t *buf1,
                         uint64_t *hits, uint64_t cap, uint64_t found) {
    const uint64_t halo = j->nrows - 1;
    const uint64_t start = a > halo ? a - halo : 0;
    const uint64_t width = b - start;
    const uint8_t *sym = j->symbols + start;
    /* buffers hold the previous row shifted by one: prev[s] is the score of column s-1; prev[0] = 0 (column `start`
       sees 0: the matrix edge, or a diagonal that starts on row 0 inside the span for every kept cell) */
    uint8_t *prev = buf0, *next = buf1;
    memset(prev, 0, width + 32);
    const __m256i zero = _mm256_setzero_si256();
    const __m256i limit = _mm256_set1_epi16(255);
    for (uint64_t p = 0; p < j->nrows; p++) {
        int32_t row_word;
        memcpy(&row_word, j->model + 4 * p, 4);
        const __m128i lut = _mm_set1_epi32(row_word);              /* bytes A,C,G,T repeated: pshufb index 0..3 */
        next[0] = 0;
        for (uint64_t s = 0; s < width; s += 16) {
            const __m128i sy = _mm_loadu_si128((const __m128i *)(sym + s));
            const __m256i m = _mm256_cvtepi8_epi16(_mm_shuffle_epi8(lut, sy));
            const __m256i old = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)(prev + s)));
            __m256i t = _mm256_add_epi16(old, m);
            const __m256i hit = _mm256_cmpgt_epi16(t, limit);       /* t >= 256 */
            t = _mm256_andnot_si256(hit, _mm256_max_epi16(t, zero));
            const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi16(t, t), 0x08);
            _mm_storeu_si128((__m128i *)(next + s + 1), _mm256_castsi256_si128(packed));
            const unsigned mask = (unsigned)_mm256_movemask_epi8(hit);
            if (mask) {
                for (unsigned k = 0; k < 16; k++) {
                    if (!(mask & (2u << (2 * k)))) continue;
                    const uint64_t col = start + s + k;
                    if (s + k < width && col >= a) {
                        if (found < cap) hits[found] = havac_oracle_pack_hit((uint32_t)p, col);
                        found++;
                    }
                }
            }
        }
        uint8_t *tmp = prev; prev = next; next = tmp;
    }
    return (int64_t)found;
}

__attribute__((target("avx2")))
static void *fast_main(void *arg) {
    struct fast_job *j = (struct fast_job *)arg;
    const uint64_t span = j->tile + j->nrows + 64;
    uint8_t *buf0 = (uint8_t *)malloc(span + 64), *buf1 = (uint8_t *)malloc(span + 64);
    if (!buf0 || !buf1) { free(buf0); free(buf1); j->found = -1; return NULL; }
    uint64_t found = 0;
    for (uint64_t t = j->tile_begin; t < j->tile_end; t++) {
        uint64_t a = t * j->tile, b = a + j->tile;
        if (b > j->n) b = j->n;
        found = (uint64_t)fast_tile(j, a, b, buf0, buf1, j->hits, j->cap, found);
    }
    free(buf0); free(buf1);
    j->found = (int64_t)found;
    return NULL;
}

int64_t havac_oracle_ssv_fast(const uint8_t *symbols, uint64_t n, const int8_t *model,
                              uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads) {
    if (!__builtin_cpu_supports("avx2")) return havac_oracle_ssv_mt(symbols, n, model, nrows, hits, cap, nthreads);
    if (n == 0 || nrows == 0) return 0;
    if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads < 1) nthreads = 1;
    const uint64_t tile = fast_tile_width(nrows);
    const uint64_t ntiles = (n + tile - 1) / tile;
    if ((uint64_t)nthreads > ntiles) nthreads = (int)ntiles;
    /* the symbol loads read up to 15 bytes past the span of the last tile: work on a padded copy */
    uint8_t *padded = (uint8_t *)malloc((size_t)n + 64);
    struct fast_job *jobs = (struct fast_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof *tids);
    if (!padded || !jobs || !tids) { free(padded); free(jobs); free(tids); return -1; }
    memcpy(padded, symbols, (size_t)n);
    memset(padded + n, 0, 64);
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (struct fast_job){padded, n, model, nrows, tile, ntiles * (uint64_t)t / (uint64_t)nthreads,
                                    ntiles * (uint64_t)(t + 1) / (uint64_t)nthreads, NULL, cap, 0};
        jobs[t].hits = (uint64_t *)malloc((size_t)(cap ? cap : 1) * sizeof(uint64_t));
        if (!jobs[t].hits) total = -1;
    }
    if (total == 0) {
        for (int t = 0; t < nthreads; t++) pthread_create(&tids[t], NULL, fast_main, &jobs[t]);
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
    free(padded); free(jobs); free(tids);
    return total;
}
#else
int64_t havac_oracle_ssv_fast(const uint8_t *symbols, uint64_t n, const int8_t *model,
                              uint64_t nrows, uint64_t *hits, uint64_t cap, int nthreads) {
    return havac_oracle_ssv_mt(symbols, n, model, nrows, hits, cap, nthreads);
}
#endif
