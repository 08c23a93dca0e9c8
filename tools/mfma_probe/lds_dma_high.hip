// Does an LDS-DMA (buffer_load_dword ... lds) reach LDS addresses above 64 KiB on gfx950 (160 KiB of LDS; M0 carries the base)?
// One workgroup: 64 lanes fetch 64 floats into LDS at byte offsets 0, 70,000-ish and 130,000-ish, then read them back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe/lds_dma_high tools/mfma_probe/lds_dma_high.hip && tools/mfma_probe/lds_dma_high
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const float* x, float* y, int n) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[150 * 1024];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned offs[3] = {0u, 69632u, 131072u};
    for (int i = threadIdx.x; i < 150 * 256; i += 64) reinterpret_cast<float*>(smem)[i] = -1.f;
    __syncthreads();
    const unsigned long long base = (unsigned long long)(size_t)x;
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) & 0xffffu, (unsigned)n * 4u, 0x00020000u};
    for (int t = 0; t < 3; ++t) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + offs[t]);
        const unsigned voff = (threadIdx.x == 5) ? 0x80000000u : (unsigned)(threadIdx.x + 64 * t) * 4u;  // lane 5 out of range -> 0
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs), "s"(0u) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < 3; ++t) y[t * 64 + threadIdx.x] = reinterpret_cast<float*>(smem + offs[t])[threadIdx.x];
}
int main() {
    float *x, *y, hx[192], hy[192];
    for (int i = 0; i < 192; ++i) hx[i] = 1000.f + i;
    hipMalloc(&x, sizeof(hx)); hipMalloc(&y, sizeof(hy));
    hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 192);
    hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 3; ++t) {
        for (int l = 0; l < 64; ++l) {
            const float want = l == 5 ? 0.f : 1000.f + l + 64 * t;
            if (hy[t * 64 + l] != want) { if (bad < 10) printf("offset #%d lane %d: got %g want %g\n", t, l, hy[t * 64 + l], want); ++bad; }
        }
    }
    printf("LDS-DMA to offsets 0 / 69632 / 131072: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
