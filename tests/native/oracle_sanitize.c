/* The CPU checker under AddressSanitizer / UBSan (tests/test_sanitizers.py compiles and runs this): the three routes
 * to the hit list on random problems, ragged sizes, tall models; they must agree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssv_oracle.h"

int main(void) {
    srand(3);
    for (int rep = 0; rep < 14; rep++) {
        uint64_t n = 1 + (uint64_t)rand() % 40000, nrows = 1 + (uint64_t)rand() % (rep % 7 == 0 ? 9000 : 900);
        uint8_t *sym = malloc(n);
        int8_t *model = malloc(nrows * 4);
        for (uint64_t i = 0; i < n; i++) sym[i] = (uint8_t)(rand() & 3);
        for (uint64_t i = 0; i < nrows * 4; i++)
            model[i] = (rep & 1) ? (int8_t)(rand() % 256 - 128) : (int8_t)((rand() % 4 == 0) ? 30 : -40);
        const uint64_t cap = 1 << 16;
        uint64_t *a = malloc(cap * 8), *b = malloc(cap * 8), *c = malloc(cap * 8);
        int64_t na = havac_oracle_ssv(sym, n, model, nrows, a, cap);
        int64_t nb = havac_oracle_ssv_fast(sym, n, model, nrows, b, cap, 1 + rep % 5);
        int64_t nc = havac_oracle_ssv_mt(sym, n, model, nrows, c, cap, 1 + rep % 3);
        uint64_t m = (uint64_t)(na < (int64_t)cap ? na : (int64_t)cap);
        havac_oracle_sort_device_order(a, m);
        int same = na == nb && na == nc && (na > (int64_t)cap || (memcmp(a, b, m * 8) == 0 && memcmp(a, c, m * 8) == 0));
        if (!same) { printf("MISMATCH rep %d: %ld %ld %ld\n", rep, (long)na, (long)nb, (long)nc); return 1; }
        uint8_t *packed = malloc((n + 3) / 4), *back = malloc(n);
        havac_oracle_pack_2bit(sym, n, packed);
        havac_oracle_unpack_2bit(packed, n, back);
        if (memcmp(sym, back, n)) { printf("pack mismatch\n"); return 1; }
        free(sym); free(model); free(a); free(b); free(c); free(packed); free(back);
    }
    printf("oracle sanitizer run ok\n");
    return 0;
}
