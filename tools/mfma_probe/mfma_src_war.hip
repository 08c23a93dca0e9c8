// Is an MFMA's A / B source register safe from a LATER instruction's write while the MFMA waits for its C operand?
//
// A wave issues a chain of DEPTH dependent v_mfma_f32_32x32x16_f16 (each takes the previous one's result as C) on fragments (A, B) read
// from LDS, and right behind the chain overwrites A and B - by ds_read_b128 from another LDS region.
// If the hardware reads an MFMA's A / B when the instruction is ISSUED, the chain's result is (DEPTH x A.B); if it reads them when the
// MFMA STARTS (behind the chain's earlier links and behind whatever the other wave of the SIMD has in the matrix pipe), the later links
// see the new fragments.  Two workgroups per CU (launch bounds (256, 2)), the second one a matrix-pipe hog, make the wait long.
// Every lane checks its own sixteen results exactly (integers in fp16): any other value is counted.
// hipcc --offload-arch=gfx950 -O3 -o mfma_src_war mfma_src_war.hip ; ./mfma_src_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int DEPTH, int MODE>
__global__ void __launch_bounds__(256, 2) probe(unsigned long long* wrong, unsigned long long* checked, int iters, int hog_mfmas) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][64][8];
    const int lane = threadIdx.x & 63;
    if ((blockIdx.x >> 8) & 1) {  // the hog (blocks 256-511, 768-1023, ...: dealt onto the CUs of blocks 0-255, 512-767, ... as their second workgroup): nothing but independent MFMAs
        f32x16 c[4] = {};
        h8 a, b;
        for (int k = 0; k < 8; ++k) a[k] = b[k] = (_Float16)(1 + (lane & 1));
        for (int i = 0; i < iters * hog_mfmas / 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
        if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) wrong[1] = 1;
        return;
    }
    // region 0: all ones (A . B over K = 16: 16); region 1: all twos (64 per product)
    for (int k = 0; k < 8; ++k) {
        lds[0][lane][k] = (_Float16)1;
        lds[1][lane][k] = (_Float16)2;
    }
    __syncthreads();
    unsigned long long bad = 0, n = 0;
    for (int it = 0; it < iters; ++it) {
        h8 a = *reinterpret_cast<const h8*>(&lds[0][lane][0]);
        h8 b = *reinterpret_cast<const h8*>(&lds[0][(lane + 1) & 63][0]);
        f32x16 c = {};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        // overwrite the fragments right behind the chain
        // (inline asm with the fragment as a read-write operand: the SAME physical registers; the compiler renames a plain assignment)
        if (MODE == 0) {
            const unsigned aa = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)&lds[1][lane][0];
            const unsigned ab = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)&lds[1][(lane + 1) & 63][0];
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "+v"(a), "+v"(b) : "v"(aa), "v"(ab) : "memory");
        }
        // keep the new fragments alive (they feed a second accumulator that is checked too: 64 per link)
        f32x16 c2 = {};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bad += c[r] != 16.f * DEPTH;
            bad += MODE == 0 && c2[r] != 64.f;
            n += 2;
        }
    }
    atomicAdd(wrong, bad);
    atomicAdd(checked, n);
}

template <int DEPTH, int MODE>
static void run(const char* what, unsigned long long* d, int hog) {
    CK(hipMemset(d, 0, 16));
    hipLaunchKernelGGL((probe<DEPTH, MODE>), dim3(2048), dim3(256), 0, 0, d, d + 1, 2000, hog);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("chain of %d, fragments overwritten by %-12s, hog %2d MFMAs per iteration: %llu wrong of %llu results\n", DEPTH, what, hog, h[0], h[1]);
}

int main() {
    unsigned long long* d;
    CK(hipMalloc(&d, 16));
    for (int hog : {0, 8, 32}) {
        run<1, 0>("ds_read_b128", d, hog);
        run<3, 0>("ds_read_b128", d, hog);
        run<6, 0>("ds_read_b128", d, hog);
    }
    return 0;
}
