// lds_lookup_bench.hip -- can LDS serve part of the match-score lookups while the VALU does the adds?
// Per row PAIR a wave does, for each of its 16 score registers, 2 lookups + 2 v_pk_add_i16 clamp, plus
// 8 v_or3 and one wave-wide test (the shape of ssv_diag_kernel's row_pair_step).  K of the 16 registers
// take both rows' match words from ONE ds_read_b64 (a 64-entry x 8 B table per row pair, indexed by a
// 3-symbol code); the other 16-K use two v_perm_b32.  Reports cycles per row per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sadd(uint32_t a, uint32_t b) {
    short2v r = __builtin_elementwise_add_sat(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(uint32_t, r);
}

constexpr int PAIRS = 16;      // row pairs per pass (one 32-row chunk of tables resident: 8 KB per wave)
constexpr int ITERS = 512;
#ifndef CODES
#define CODES 64u
#endif

template <int K>
__global__ __launch_bounds__(256, 4) void k(uint32_t* out, const uint32_t* codes_in, const uint32_t* rows) {
    __shared__ uint2 tab[4][PAIRS * 64];         // 8 KB per wave
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x & 63; i < PAIRS * 64; i += 64) tab[wave][i] = make_uint2(i * 2654435761u & 0xff00ff00u, i * 40503u & 0xff00ff00u);
    __syncthreads();
    uint32_t x[16], code[16], sel[16];
    for (int i = 0; i < 16; i++) {
        uint32_t c = codes_in[(blockIdx.x * 256 + threadIdx.x) * 16 + i];
        x[i] = 0x80008000u;
        code[i] = (c & (CODES - 1)) * 8u;
        sel[i] = 0x000c000cu | ((c & 3u) << 8) | (((c >> 2) & 3u) << 24);
    }
    const char* base = (const char*)tab[wave];
    uint32_t acc = 0;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int p = 0; p < PAIRS; p++) {
            asm volatile("" ::: "memory");
            const uint32_t r0 = rows[2 * p], r1 = rows[2 * p + 1];     // scalar loads (uniform)
            uint32_t m0[16], m1[16], y[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i < K) {
                    uint2 v = *(const uint2*)(base + p * 512 + code[i]);
                    m0[i] = v.x; m1[i] = v.y;
                } else {
                    m0[i] = __builtin_amdgcn_perm(r0, r0, sel[i]);
                    m1[i] = __builtin_amdgcn_perm(r1, r1, sel[i]);
                }
            }
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = sadd(x[i], m0[i]);
#pragma unroll
            for (int i = 0; i < 16; i++) { x[i] = sadd(y[i], m1[i]); any |= x[i]; }
            if (__builtin_expect(__any((any & 0x00010001u) != 0), 0)) { x[0] = 0x80008000u; acc++; }
        }
    }
    uint32_t r = acc;
    for (int i = 0; i < 16; i++) r ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int K>
void run(uint32_t* out, const uint32_t* codes, const uint32_t* rows, int blocks, double ghz, int wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<K><<<blocks, 256>>>(out, codes, rows);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<K><<<blocks, 256>>>(out, codes, rows);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double cycles = ms * 1e-3 * ghz * 1e9;
    double per_row = cycles / ((double)ITERS * PAIRS * 2 * wps);
    printf("K=%2d of 16 registers via LDS: %8.3f ms  %6.1f cycles per wave-row per SIMD (%d waves/SIMD)\n", K, ms, per_row, wps);
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    double ghz = p.clockRate / 1e6;
    int blocks = p.multiProcessorCount * wps;
    uint32_t *out, *codes, *rows;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&codes, (size_t)blocks * 256 * 16 * 4);
    (void)hipMalloc(&rows, 64 * 4);
    uint32_t* h = (uint32_t*)malloc((size_t)blocks * 256 * 16 * 4);
    for (size_t i = 0; i < (size_t)blocks * 256 * 16; i++) h[i] = (uint32_t)rand();
    (void)hipMemcpy(codes, h, (size_t)blocks * 256 * 16 * 4, hipMemcpyHostToDevice);
    uint32_t hr[64]; for (int i = 0; i < 64; i++) hr[i] = 0xc8d0e0f0u;   // all-negative rows: no crossing
    (void)hipMemcpy(rows, hr, sizeof hr, hipMemcpyHostToDevice);
    printf("%s %d CUs %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
    run<0>(out, codes, rows, blocks, ghz, wps);
    run<6>(out, codes, rows, blocks, ghz, wps);
    run<8>(out, codes, rows, blocks, ghz, wps);
    run<10>(out, codes, rows, blocks, ghz, wps);
    run<12>(out, codes, rows, blocks, ghz, wps);
    run<14>(out, codes, rows, blocks, ghz, wps);
    run<16>(out, codes, rows, blocks, ghz, wps);
    return 0;
}
