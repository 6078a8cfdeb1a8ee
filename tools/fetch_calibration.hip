// fetch_calibration.hip -- what do FETCH_SIZE / WRITE_SIZE count on gfx950 for the access widths the SSV kernel uses?
// Streams a 1 GiB buffer (far beyond L2 and the Infinity Cache) once per kernel with 4-, 8- and 16-byte-per-lane loads and
// writes it once with 8-byte-per-lane stores; tools/fetch_calibration.sh runs it under rocprofv3 --pmc and prints
// counter x 1024 / bytes really moved.  (bench.py's roofline.traffic uses 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950
// correction; the SSV kernel's loads are 8 bytes per lane.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fetch_calibration tools/fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__device__ __forceinline__ void stream_read(const T* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x9e3779b9u) out[0] = acc;      // never true for a zeroed buffer: keeps the loads alive
}
__global__ void stream_read4(const uint32_t* p, size_t n, uint32_t* out) { stream_read<uint32_t>(p, n, out); }
__global__ void stream_read8(const uint2* p, size_t n, uint32_t* out) { stream_read<uint2>(p, n, out); }
__global__ void stream_read16(const uint4* p, size_t n, uint32_t* out) { stream_read<uint4>(p, n, out); }
__global__ void stream_write8(uint2* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint2((uint32_t)i, 1u);
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf = nullptr; uint32_t* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(stream_read4, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(stream_read8, grid, block, 0, 0, (const uint2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(stream_read16, grid, block, 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(stream_write8, grid, block, 0, 0, (uint2*)buf, bytes / 8);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernels failed\n"); return 1; }
    std::printf("streamed %zu bytes per kernel\n", bytes);
    return 0;
}
