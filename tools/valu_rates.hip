// valu_rates.hip -- how many cycles a wave64 VALU instruction occupies a gfx950 SIMD, per opcode.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/valu_rates.hip ; run on the GPU box.
// Each kernel runs ITER x 16 independent copies of one instruction per wave, 4 waves per SIMD on every
// SIMD of the chip; cycles/instr = elapsed * clock / (ITER*16*4).  Results: profiles/valu_rates_r01.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEFK(name, ASM)                                                                     \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t a, uint32_t b) {  \
        uint32_t v[16];                                                                     \
        for (int i = 0; i < 16; i++) v[i] = threadIdx.x * (i + 1) + a;                      \
        uint32_t s = b + threadIdx.x, t = a ^ threadIdx.x;                                  \
        for (int it = 0; it < ITER; it++) {                                                 \
            REP16(ASM)                                                                      \
        }                                                                                   \
        uint32_t r = 0;                                                                     \
        for (int i = 0; i < 16; i++) r ^= v[i];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                     \
    }

#define A_PERM(i)   asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_PKADD(i)  asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(s));
#define A_PKADDN(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_PKADDU(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_PKMAX(i)  asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_OR3(i)    asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_ADD(i)    asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_AND(i)    asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_OR(i)     asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_ADD3(i)   asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_ALIGN(i)  asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(v[i]) : "v"(s));
#define A_BFE(i)    asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(v[i]));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(s));
#define A_MAD24(i)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_MUL24(i)  asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_MULLO(i)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_ADD16(i)  asm volatile("v_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_FMA(i)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_FMAC(i)   asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_ADDF(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_PKFMA(i)  asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_MOV(i)    asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(s));
#define A_XOR(i)    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_MAX16(i)  asm volatile("v_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(s));
#define A_SADU8(i)  asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_DOT4(i)   asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_CNDM(i)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(s));
#define A_BFI(i)    asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(s), "v"(t));
#define A_PERMS(i)  asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(v[i]) : "s"(a), "v"(t));

DEFK(perm, A_PERM) DEFK(perm_sgpr, A_PERMS) DEFK(pk_add_i16_clamp, A_PKADD) DEFK(pk_add_i16, A_PKADDN) DEFK(pk_add_u16, A_PKADDU)
DEFK(pk_max_i16, A_PKMAX) DEFK(or3, A_OR3) DEFK(add_u32, A_ADD) DEFK(and_b32, A_AND) DEFK(or_b32, A_OR) DEFK(add3, A_ADD3)
DEFK(alignbit, A_ALIGN) DEFK(bfe, A_BFE) DEFK(lshl_or, A_LSHLOR) DEFK(mad_u24, A_MAD24) DEFK(mul_u24, A_MUL24)
DEFK(mul_lo_u32, A_MULLO) DEFK(add_u16, A_ADD16) DEFK(fma_f32, A_FMA) DEFK(fmac_f32, A_FMAC) DEFK(add_f32, A_ADDF)
DEFK(pk_fma_f16, A_PKFMA) DEFK(mov, A_MOV) DEFK(xor_b32, A_XOR) DEFK(max_i16, A_MAX16) DEFK(sad_u8, A_SADU8)
DEFK(dot4_i32_i8, A_DOT4) DEFK(cndmask, A_CNDM) DEFK(bfi, A_BFI)

template <typename K>
void run(const char* name, K kern, uint32_t* out, int blocks, double clock_ghz, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 1, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    double cycles = ms * 1e-3 * clock_ghz * 1e9;
    double per = cycles / ((double)ITER * 16 * waves_per_simd);
    printf("%-20s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at %.2f GHz, %d waves/SIMD)\n", name, ms, per,
           clock_ghz, waves_per_simd);
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;   // waves per SIMD
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, clock %.3f GHz\n", p.gcnArchName, cus, ghz);
    int blocks = cus * wps;   // one 256-thread block = 1 wave per SIMD of a CU
    uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
#define RUN(n) run(#n, k_##n, out, blocks, ghz, wps);
    RUN(perm) RUN(perm_sgpr) RUN(pk_add_i16_clamp) RUN(pk_add_i16) RUN(pk_add_u16) RUN(pk_max_i16) RUN(or3) RUN(add_u32) RUN(and_b32)
    RUN(or_b32) RUN(xor_b32) RUN(mov) RUN(add3) RUN(alignbit) RUN(bfe) RUN(lshl_or) RUN(bfi) RUN(cndmask) RUN(mad_u24) RUN(mul_u24)
    RUN(mul_lo_u32) RUN(add_u16) RUN(max_i16) RUN(sad_u8) RUN(dot4_i32_i8) RUN(fma_f32) RUN(fmac_f32) RUN(add_f32) RUN(pk_fma_f16)
    return 0;
}
