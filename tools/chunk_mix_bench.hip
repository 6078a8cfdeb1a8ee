// chunk_mix_bench.hip -- what the chip does with the hot loop's instruction mix and NOTHING ELSE: per window of four steps
// 4 x (8 conflict-free ds_read_b64 + 16 v_pk_add_i16 clamp), the 17-input OR + mask (7 v_or3_b32 + v_bitop3_b32), one compare
// and one branch; eight windows per "chunk"; no global loads, no table build, no window expansion, no slow path.  Six waves per
// SIMD on every SIMD of the chip (W = waves per SIMD: 4 .. 8 by the launch bounds), tables and addresses as in ssv_diag_kernel
// (16 step-pair tables of 17 x 8 B per wave, entry = code * 8).  Reports SIMD-cycles per chunk and wave -- ssv_diag_kernel's
// chunk takes 2,884 with hits, 2,717 without (C2), of which 2,438 are the vector ALU's nominal issue cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/chunk_mix_bench tools/chunk_mix_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_words_t;
constexpr int kPairStride = 17 * 8, kTableBytes = 2304, kChunks = 256;

template <int READS /* 1: the match words come from LDS; 0: the adds use a register (the vector ALU alone) */, int ORS /* 1: with the OR tree and the test */>
__device__ __forceinline__ void chunk(uint32_t (&a)[16], uint32_t (&b)[16], const uint32_t (&C)[32], uint32_t mask, uint32_t& hits, uint32_t filler) {
#pragma unroll
    for (int Q = 0; Q < 8; Q++) {
        uint32_t (&cur)[16] = (Q & 1) ? b : a;
        uint32_t (&nxt)[16] = (Q & 1) ? a : b;
#pragma unroll
        for (int half = 0; half < 2; half++) {      // steps 4Q, 4Q+1 then 4Q+2, 4Q+3
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u32x2 m[8];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if constexpr (READS) m[i] = *(lds_words_t)(uintptr_t)(C[2 * Q + half + h * 8 + i] + (2 * Q + half) * kPairStride);
                    else m[i] = u32x2{filler, filler};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (half == 0) asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(nxt[h * 8 + i]) : "v"(cur[h * 8 + i]), "v"(m[i].x));
                    else asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[h * 8 + i]) : "v"(m[i].x));
                }
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[h * 8 + i]) : "v"(m[i].y));
            }
        }
        if constexpr (ORS) {
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) any |= nxt[i];
            if (__builtin_expect(__ballot((any & mask) != 0) != 0, 0)) { hits++; nxt[0] = 0x80008000u; }
        }
    }
}

// the same bytes from LDS in HALF as many instructions: a window's four match words of a register in one ds_read_b128 (a
// table such reads could index does not fit LDS -- 4 KB per window and wave; this only asks whether a read's cost to the
// vector issue goes by the instruction or by the byte)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_quad_t;
template <int ORS>
__device__ __forceinline__ void chunk_b128(uint32_t (&a)[16], uint32_t (&b)[16], const uint32_t (&C)[32], uint32_t mask, uint32_t& hits) {
#pragma unroll
    for (int Q = 0; Q < 8; Q++) {
        uint32_t (&cur)[16] = (Q & 1) ? b : a;
        uint32_t (&nxt)[16] = (Q & 1) ? a : b;
#pragma unroll
        for (int g = 0; g < 4; g++) {               // four registers at a time: 4 reads of 16 B, 16 adds
            u32x4 m[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = *(lds_quad_t)(uintptr_t)(((C[2 * Q + g * 4 + i] + (2 * Q) * kPairStride) & ~15u));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(nxt[g * 4 + i]) : "v"(cur[g * 4 + i]), "v"(m[i].x));
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[g * 4 + i]) : "v"(m[i].y));
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[g * 4 + i]) : "v"(m[i].z));
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(nxt[g * 4 + i]) : "v"(m[i].w));
        }
        if constexpr (ORS) {
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) any |= nxt[i];
            if (__builtin_expect(__ballot((any & mask) != 0) != 0, 0)) { hits++; nxt[0] = 0x80008000u; }
        }
    }
}

template <int W, int READS, int ORS>
__global__ __launch_bounds__(256, W) void k(uint32_t* out, const uint32_t* codes, uint32_t mask, uint32_t filler, uint64_t* clocks) {
    __shared__ __attribute__((aligned(128))) uint8_t tables[4][kTableBytes + 1024];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = lane; i < kTableBytes / 4; i += 64) reinterpret_cast<uint32_t*>(tables[wave])[i] = 0xf000f000u;      // every match word: -16, -16 (no cell ever leaves score 0)
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tables[wave];
    uint32_t C[32];
#pragma unroll
    for (int i = 0; i < 32; i++) C[i] = base + (codes[(blockIdx.x * 256 + threadIdx.x) * 32 + i] & 15u) * 8u;
    uint32_t a[16], b[16], hits = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = b[i] = 0x80008000u;
    const uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();      // shader-clock cycles (s_memtime) and the 100 MHz clock
#pragma unroll 1
    for (int c = 0; c < kChunks; c++) {
        if constexpr (READS == 2) chunk_b128<ORS>(a, b, C, mask, hits);
        else chunk<READS, ORS>(a, b, C, mask, hits, filler);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = __builtin_readcyclecounter() - c0; clocks[1] = wall_clock64() - w0; }
    uint32_t r = hits;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= a[i] ^ b[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int W, int READS, int ORS>
static void run(const char* what, uint32_t* out, const uint32_t* codes, int cus, double ghz) {
    static uint64_t* clocks = nullptr;
    if (!clocks) (void)hipHostMalloc(&clocks, 16, hipHostMallocDefault);
    const int blocks = cus * W;       // W waves per SIMD: W workgroups of four waves per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, READS, ORS>), dim3(blocks), dim3(256), 0, 0, out, codes, 0x00010001u, 0xf000f000u, clocks);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // a SIMD holds W waves, each does kChunks chunks: SIMD-cycles per chunk and wave = elapsed cycles / (W * kChunks)
    const double cycles = ms * 1e-3 * ghz * 1e9 / (W * (double)kChunks);
    // the first wave's own clocks around its loop: s_memtime ticks per 100 MHz tick = what the shader clock really was
    const double real_ghz = clocks[1] ? (double)clocks[0] / (double)clocks[1] * 0.1 : 0.0;
    std::printf("%-64s %d waves/SIMD  %8.3f ms  %7.1f SIMD-cycles per chunk and wave at %.2f GHz;  s_memtime / s_memrealtime of one wave: %.3f GHz -> %7.1f\n",
                what, W, ms, cycles, ghz, real_ghz, real_ghz > 0 ? ms * 1e-3 * real_ghz * 1e9 / (W * (double)kChunks) : 0.0);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { std::fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    std::printf("%s: %d CUs, %.2f GHz; nominal issue cycles per chunk: 512 v_pk_add_i16 x 4 = 2048, + 64 three-input ORs x 4 + 8 compares x 2 = 2320\n", prop.gcnArchName, cus, ghz);
    const size_t threads = (size_t)cus * 8 * 256;
    uint32_t *out = nullptr, *codes = nullptr;
    (void)hipMalloc(&out, threads * 4);
    (void)hipMalloc(&codes, threads * 32 * 4);
    uint32_t* h = (uint32_t*)std::malloc(threads * 32 * 4);
    for (size_t i = 0; i < threads * 32; i++) h[i] = (uint32_t)std::rand();
    (void)hipMemcpy(codes, h, threads * 32 * 4, hipMemcpyHostToDevice);
    run<6, 0, 0>("adds alone (match words in a register)", out, codes, cus, ghz);
    run<6, 0, 1>("adds + OR tree + test", out, codes, cus, ghz);
    run<6, 1, 0>("adds + 256 ds_read_b64", out, codes, cus, ghz);
    run<6, 1, 1>("adds + reads + OR tree + test (the chunk's mix)", out, codes, cus, ghz);
    run<6, 2, 0>("adds + 128 ds_read_b128 (the same bytes, half the reads)", out, codes, cus, ghz);
    run<6, 2, 1>("adds + 128 ds_read_b128 + OR tree + test", out, codes, cus, ghz);
    run<4, 1, 1>("adds + reads + OR tree + test", out, codes, cus, ghz);
    run<5, 1, 1>("adds + reads + OR tree + test", out, codes, cus, ghz);
    std::free(h);
    return 0;
}
