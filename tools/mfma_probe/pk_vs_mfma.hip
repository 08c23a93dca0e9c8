// Does a packed fp32 vector instruction lose results while ANOTHER wave of the SIMD issues MFMAs?  (profiles/probes_r05.md section 4: the
// root cause of round 4's Gram fault, here without the Gram kernel.)
//
// Workgroups 0-255, 512-767, ... ("victims") run nothing but dependent v_pk_fma_f32 / v_fma_f32 chains with exactly representable results
// (x <- x * 1 + 1: after n steps x = n + 1 in every lane, both halves of the pair); workgroups 256-511, 768-1023, ... - dealt onto the same
// CUs as their second workgroup (70 KB of LDS each: two per CU) - run a stand-in: MFMAs only (mode 1), nothing (mode 0).  Every victim lane
// checks its own results; wrong lanes are counted per (register half, lane quarter).
// hipcc --offload-arch=gfx950 -O3 -o pk_vs_mfma pk_vs_mfma.hip ; ./pk_vs_mfma [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int PACKED>
__global__ void __launch_bounds__(256, 2) probe(unsigned long long* wrong, int steps, int hog_iters, int hog) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    if ((blockIdx.x >> 8) & 1) {
        if (!hog) return;
        f32x16 c[4] = {};
        h8 a, b;
        for (int k = 0; k < 8; ++k) a[k] = b[k] = (_Float16)(1 + (lane & 1));
        for (int i = 0; i < hog_iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
        if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) wrong[15] = 1;
        return;
    }
    // sixteen independent chains (register pairs), so that the vector pipe is busy like a fold of accumulators
    f32x2 x[16];
    const f32x2 one = {1.f, 1.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = one;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (PACKED) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[r]) : "v"(one));
            } else {
                asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(x[r][0]), "+v"(x[r][1]) : "v"(1.f));
            }
        }
    }
    unsigned long long bad_lo = 0, bad_hi = 0;
    const float want = (float)(steps + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bad_lo += x[r][0] != want;
        bad_hi += x[r][1] != want;
    }
    if (bad_lo) atomicAdd(&wrong[lane >> 4], bad_lo);          // low register of the pair, by lane quarter
    if (bad_hi) atomicAdd(&wrong[4 + (lane >> 4)], bad_hi);    // high register
    if (threadIdx.x == 0) atomicAdd(&wrong[8], 1ull);          // victims that ran
    lds[threadIdx.x] = x[0][0];
}

// The fold of an MFMA accumulator, as the Gram kernel does it: the victim multiplies (four 32 x 32 x 16 MFMAs of ones into one accumulator:
// every element 64), then - behind `gap` idle s_sleep units - adds the accumulator into a master with packed or plain FMAs and zeroes it.
template <int PACKED>
__global__ void __launch_bounds__(256, 2) probe_fold(unsigned long long* wrong, int steps, int hog_iters, int hog, int gap) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
#ifdef FULL_VGPRS  // the kernel allocates all 256 registers: the two waves of a SIMD then fill its register file, as the Gram kernel's do
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
    h8 a, b;
    for (int k = 0; k < 8; ++k) a[k] = b[k] = (_Float16)1;
    if ((blockIdx.x >> 8) & 1) {
        if (!hog) return;
        f32x16 c[4] = {};
        for (int i = 0; i < hog_iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
        if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) wrong[15] = 1;
        return;
    }
    f32x16 acc[2] = {}, master[2] = {};
    const f32x2 one = {1.f, 1.f};
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (gap) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 m2 = {master[j][r], master[j][r + 1]};
                f32x2 a2 = {acc[j][r], acc[j][r + 1]};
                if (PACKED) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m2) : "v"(a2), "v"(one));
                } else {
                    asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(m2[0]), "+v"(m2[1]) : "v"(a2[0]), "v"(a2[1]), "v"(1.f));
                }
                master[j][r] = m2[0];
                master[j][r + 1] = m2[1];
                acc[j][r] = acc[j][r + 1] = 0.f;
            }
    }
    unsigned long long bad_lo = 0, bad_hi = 0;
    const float want = 64.f * steps;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bad_lo += master[j][r] != want;
            bad_hi += master[j][r + 1] != want;
        }
    if (bad_lo) atomicAdd(&wrong[lane >> 4], bad_lo);
    if (bad_hi) atomicAdd(&wrong[4 + (lane >> 4)], bad_hi);
    if (threadIdx.x == 0) atomicAdd(&wrong[8], 1ull);
    lds[threadIdx.x] = master[0][0];
}

template <int PACKED>
static void run_fold(const char* what, unsigned long long* d, int launches, int hog, int gap) {
    CK(hipMemset(d, 0, 16 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_fold<PACKED>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    for (int it = 0; it < launches; ++it) hipLaunchKernelGGL((probe_fold<PACKED>), dim3(1024), dim3(256), 70 * 1024, 0, d, 400, 6000, hog, gap);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("fold of an MFMA accumulator by %-13s beside %-8s%s: wrong, low register by lane quarter %llu %llu %llu %llu, high register %llu %llu %llu %llu  (%llu victim workgroups)\n", what,
           hog ? "MFMAs" : "nothing", gap ? ", s_sleep before the fold" : "", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
}

template <int PACKED>
static void run(const char* what, unsigned long long* d, int launches, int hog) {
    CK(hipMemset(d, 0, 16 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PACKED>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    for (int it = 0; it < launches; ++it) hipLaunchKernelGGL((probe<PACKED>), dim3(1024), dim3(256), 70 * 1024, 0, d, 2000, 6000, hog);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("%-13s beside %-10s: wrong results, low register by lane quarter %llu %llu %llu %llu, high register %llu %llu %llu %llu  (%llu victim workgroups x 256 lanes x 32 results)\n", what,
           hog ? "MFMAs" : "nothing", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    unsigned long long* d;
    CK(hipMalloc(&d, 16 * 8));
    run<1>("v_pk_fma_f32", d, launches, 1);
    run<1>("v_pk_fma_f32", d, launches, 0);
    run<0>("v_fma_f32", d, launches, 1);
    for (int gap = 0; gap < 2; ++gap) {
        run_fold<1>("v_pk_fma_f32", d, launches, 1, gap);
        run_fold<1>("v_pk_fma_f32", d, launches, 0, gap);
        run_fold<0>("v_fma_f32", d, launches, 1, gap);
    }
    return 0;
}
