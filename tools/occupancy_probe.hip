// occupancy_probe.hip -- how many workgroups of each SSV kernel the runtime places on one compute unit (registers, LDS), and the
// device's LDS per CU.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I havac_amd/csrc -o tools/_bin/occupancy_probe tools/occupancy_probe.hip
#include "ssv_kernels.hip.h"
#include <cstdio>
using namespace havac;
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerMultiprocessor %d\n", p.gcnArchName,
           p.multiProcessorCount, p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor);
    int n = 0;
    hipFuncAttributes a;
#define SHOW(k) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, 0); (void)hipFuncGetAttributes(&a, (const void*)k); \
    printf("%-26s %d workgroups per CU (static LDS %zu B, %d registers)\n", #k, n, a.sharedSizeBytes, a.numRegs);
    SHOW(ssv_diag_kernel) SHOW(ssv_resident_kernel) SHOW(ssv_diag_kernel_traced)
    return 0;
}
