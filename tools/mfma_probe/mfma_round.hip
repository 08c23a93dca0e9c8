// How does v_mfma_f32_32x32x16_bf16 round the sum of its 16 products + C?  (RNE, toward zero, or toward -inf?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__global__ void k(const float* av, const float* bv, float c0, float* out) {
    // lane l: A[row l&31][k = 8*(l>>5)+j], B[k][col l&31]; use row 0 / col 0 only: lanes 0 and 32 carry k = 0..15
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * (lane >> 5) + j;
        a[j] = (__bf16)((lane & 31) == 0 ? av[kk] : 0.f);
        b[j] = (__bf16)((lane & 31) == 0 ? bv[kk] : 0.f);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];  // D[row 0][col 0]
}
static float bf(float x) { unsigned u; memcpy(&u, &x, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; float y; memcpy(&y, &u, 4); return y; }
int main() {
    float *da, *db, *dout; hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    const char* names[] = {"one big + 15 small positive", "same, all negated", "mixed signs", "C big, products small positive", "C big negative, products small negative"};
    for (int t = 0; t < 5; ++t) {
        float a[16], b[16], c0 = 0.f;
        for (int i = 0; i < 16; ++i) { a[i] = bf(1.0f + i * 0.0078125f); b[i] = bf((i == 0 ? 1.0f : 1.0f / 4096.f) * (1.0f + (i * 7 % 16) * 0.0078125f)); }
        if (t == 1) for (int i = 0; i < 16; ++i) a[i] = -a[i];
        if (t == 2) for (int i = 0; i < 16; ++i) if (i & 1) a[i] = -a[i];
        if (t == 3) { c0 = 1024.f; for (int i = 0; i < 16; ++i) b[i] = bf(b[i] * (i == 0 ? 1.f / 4096.f : 1.f)); }
        if (t == 4) { c0 = -1024.f; for (int i = 0; i < 16; ++i) { b[i] = bf(b[i] * (i == 0 ? 1.f / 4096.f : 1.f)); a[i] = -a[i]; } }
        double exact = c0; for (int i = 0; i < 16; ++i) exact += (double)a[i] * (double)b[i];
        hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
        float got; hipMemcpy(&got, dout, 4, hipMemcpyDeviceToHost);
        float rne = (float)exact;
        printf("%-45s exact %.10e  mfma %.10e  rne %.10e  (mfma-exact)/ulp %+.3f\n", names[t], exact, got, rne,
               (got - exact) / (double)(nextafterf(fabsf(rne), INFINITY) - fabsf(rne)));
    }
    return 0;
}
