// init_probe.cpp -- where the "build" time of a Havac object goes on the GPU box: HIP start-up, the code object, streams,
// the hit buffer.   hipcc -O2 tools/init_probe.cpp -o build/init_probe -Iinclude -Lhavac_amd -lhavac_dev -Wl,-rpath,$PWD/havac_amd
// and run ./build/init_probe
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "havac_dev.h"

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    double t = now(), t0 = t;
    auto lap = [&](const char* what) { double n = now(); std::printf("%-44s %8.2f ms\n", what, n - t); t = n; };
    int n = 0;
    hipGetDeviceCount(&n); lap("hipGetDeviceCount (runtime start-up)");
    hipSetDevice(0); lap("hipSetDevice");
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
    void* p = nullptr;
    hipMalloc(&p, 8); lap("first hipMalloc (8 B; context)");
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreate");
    void* big = nullptr;
    hipMalloc(&big, 14ull * 256 * 1024 * 1024); lap("hipMalloc 3.5 GiB");
    hipFree(big); lap("hipFree 3.5 GiB");
    hipMalloc(&big, 32ull << 20); lap("hipMalloc 32 MiB");
    hipMalloc(&big, 256ull << 20); lap("hipMalloc 256 MiB");
    void* h = nullptr;
    hipHostMalloc(&h, 64 << 20, hipHostMallocDefault); lap("hipHostMalloc 64 MiB pinned");
    hipEvent_t e; hipEventCreate(&e); lap("hipEventCreate");
    havac_dev* d = nullptr;
    havac_dev_create(0, &d); lap("havac_dev_create (after all of the above)");
    havac_dev_destroy(d); lap("havac_dev_destroy");
    std::printf("total %.2f ms\n", now() - t0);
    return 0;
}
