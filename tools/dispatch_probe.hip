// dispatch_probe.hip -- what does it cost to hand out the workgroups of a short-model launch?  Kernels that do (almost) nothing,
// with the SSV kernel's footprint (256 or 512 threads, 14 / 28 KB of LDS), on the grid of a one-chunk-tile launch over 100 Mbp
// (12,213 workgroups of four waves) and with twice / half / a quarter as many waves per workgroup; and the same with 3.4 us of
// sleep per wave.   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/dispatch_probe tools/dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int THREADS, int LDS_BYTES, int SLEEPS>
__global__ __launch_bounds__(THREADS) void probe(unsigned* out) {
    __shared__ unsigned lds[LDS_BYTES / 4];
    lds[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < SLEEPS; i++) __builtin_amdgcn_s_sleep(127);          // 127 x 64 cycles = 3.4 us at 2.4 GHz
    if (lds[(threadIdx.x + 1) % THREADS] == 0xdeadbeefu) out[0] = 1;
}

template <typename K>
static float run(K kernel, int blocks, int threads, unsigned* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 10 * 1000;
}

int main() {
    unsigned* out = nullptr;
    hipMalloc(&out, 64);
    const int waves = 48852;
    std::printf("workgroups x threads, LDS            empty kernel      with 3.4 us of sleep per wave      (us per launch, 10 launches back to back)\n");
    std::printf("%6d x 256, 14 KB              %10.1f %20.1f\n", waves / 4, run(probe<256, 14336, 0>, waves / 4, 256, out), run(probe<256, 14336, 1>, waves / 4, 256, out));
    std::printf("%6d x 512, 28 KB              %10.1f %20.1f\n", waves / 8, run(probe<512, 28672, 0>, waves / 8, 512, out), run(probe<512, 28672, 1>, waves / 8, 512, out));
    std::printf("%6d x 128,  7 KB              %10.1f %20.1f\n", waves / 2, run(probe<128, 7168, 0>, waves / 2, 128, out), run(probe<128, 7168, 1>, waves / 2, 128, out));
    std::printf("%6d x  64,  4 KB              %10.1f %20.1f\n", waves, run(probe<64, 4096, 0>, waves, 64, out), run(probe<64, 4096, 1>, waves, 64, out));
    return 0;
}
