// Sustained rate of the bf16 MFMA shapes on random data (one wave per SIMD, operands in registers, inline asm so the
// compiler cannot reshuffle accumulators): cycles per instruction (s_memtime), in-kernel clock (s_memrealtime) and the
// chip-wide TFLOP/s they imply.  Shapes: 32x32x16, 32x32x8_1k (legacy K=8), 16x16x32, 16x16x16_1k.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define M32(OP, A, B, C) asm volatile(OP " %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B))
template <int SHAPE>
__global__ void __launch_bounds__(256) k(const float* src, float* out, unsigned long long* st, int iters, int zero) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    s16x4 a4, b4;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(zero ? 0.f : src[(lane * 8 + j) & 4095]);
        b[j] = (__bf16)(zero ? 0.f : src[(lane * 8 + j + 777) & 4095]);
    }
    for (int j = 0; j < 4; ++j) { a4[j] = __builtin_bit_cast(short, a[j]); b4[j] = __builtin_bit_cast(short, b[j]); }
    if (SHAPE == 4) {  // real fp16 values with full 11-bit mantissas
        for (int j = 0; j < 8; ++j) {
            const _Float16 ha = (_Float16)(zero ? 0.f : src[(lane * 8 + j) & 4095] * 37.f), hb = (_Float16)(zero ? 0.f : src[(lane * 8 + j + 777) & 4095] * 3.f);
            a[j] = __builtin_bit_cast(__bf16, ha);
            b[j] = __builtin_bit_cast(__bf16, hb);
        }
    }
    f32x16 A0 = {0}, A1 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0}, c6 = {0}, c7 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (SHAPE == 0) {
            M32("v_mfma_f32_32x32x16_bf16", a, b, A0); M32("v_mfma_f32_32x32x16_bf16", a, b, A1);
            M32("v_mfma_f32_32x32x16_bf16", a, b, A0); M32("v_mfma_f32_32x32x16_bf16", a, b, A1);
        } else if (SHAPE == 1) {
            M32("v_mfma_f32_32x32x8bf16_1k", a4, b4, A0); M32("v_mfma_f32_32x32x8bf16_1k", a4, b4, A1);
            M32("v_mfma_f32_32x32x8bf16_1k", a4, b4, A0); M32("v_mfma_f32_32x32x8bf16_1k", a4, b4, A1);
        } else if (SHAPE == 2) {
            M32("v_mfma_f32_16x16x32_bf16", a, b, c0); M32("v_mfma_f32_16x16x32_bf16", a, b, c1);
            M32("v_mfma_f32_16x16x32_bf16", a, b, c2); M32("v_mfma_f32_16x16x32_bf16", a, b, c3);
            M32("v_mfma_f32_16x16x32_bf16", a, b, c4); M32("v_mfma_f32_16x16x32_bf16", a, b, c5);
            M32("v_mfma_f32_16x16x32_bf16", a, b, c6); M32("v_mfma_f32_16x16x32_bf16", a, b, c7);
        } else if (SHAPE == 4) {  // fp16 operands (same bit patterns reinterpreted: random mantissas)
            M32("v_mfma_f32_32x32x16_f16", a, b, A0); M32("v_mfma_f32_32x32x16_f16", a, b, A1);
            M32("v_mfma_f32_32x32x16_f16", a, b, A0); M32("v_mfma_f32_32x32x16_f16", a, b, A1);
        } else {
            M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c0); M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c1);
            M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c2); M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c3);
            M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c4); M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c5);
            M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c6); M32("v_mfma_f32_16x16x16bf16_1k", a4, b4, c7);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += A0[r] + A1[r];
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int SHAPE>
void run(const char* name, double flop_per_inst, int per_iter, const float* src, float* out, unsigned long long* st, int zero) {
    const int iters = 300000, wgs = 256;  // one 4-wave workgroup per CU: one wave per SIMD
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k<SHAPE>, dim3(wgs), dim3(256), 0, 0, src, out, st, iters, zero);
    hipDeviceSynchronize();
    unsigned long long h[512];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, clk = 0;
    for (int i = 0; i < wgs; ++i) { cyc += (double)h[2 * i]; clk += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
    cyc /= wgs; clk /= wgs;
    const double per = cyc / ((double)iters * per_iter);
    printf("%-30s %s  %.2f cycles/inst  clock %.3f GHz  -> %.0f TFLOP/s chip-wide (1024 SIMDs)\n", name, zero ? "zeros " : "random", per,
           clk, flop_per_inst / per * clk * 1e9 * 1024 / 1e12);
}
int main() {
    float *src, *out; unsigned long long* st;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&st, 512 * 8);
    float h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int zero = 0; zero < 2; ++zero) {
        run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 4, src, out, st, zero);
        run<1>("v_mfma_f32_32x32x8bf16_1k", 2.0 * 32 * 32 * 8, 4, src, out, st, zero);
        run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 8, src, out, st, zero);
        run<3>("v_mfma_f32_16x16x16bf16_1k", 2.0 * 16 * 16 * 16, 8, src, out, st, zero);
        run<4>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 4, src, out, st, zero);
    }
    return 0;
}
