// Probe: two workgroups of one CU with more than 64 KB of dynamic LDS each - does either ever see the other's bytes?
// Every workgroup fills its whole allocation with a pattern of its own, waits, verifies, many rounds.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_isolation lds_isolation.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256, 2) probe(int words, int rounds, unsigned* bad, unsigned* cu_pairs) {
    extern __shared__ unsigned lds[];
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned pat = (blockIdx.x * 2654435761u) ^ (r * 40503u);
        for (int i = threadIdx.x; i < words; i += 256) lds[i] = pat + i;
        __syncthreads();
        __builtin_amdgcn_s_sleep(20);
        for (int i = threadIdx.x; i < words; i += 256) nbad += lds[i] != pat + i;
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    unsigned *bad;
    (void)hipMalloc(&bad, 8);
    for (int bytes : {32768, 65536, 73792, 73856, 81920, 90000}) {
        (void)hipMemset(bad, 0, 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, bytes);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), bytes, 0, bytes / 4, 200, bad, nullptr);
        hipError_t e = hipDeviceSynchronize();
        unsigned h = 0;
        (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("%6d bytes of dynamic LDS per workgroup, %d workgroups per CU by the occupancy query: %u wrong words (%s)\n", bytes, occ, h, hipGetErrorString(e));
    }
    return 0;
}
