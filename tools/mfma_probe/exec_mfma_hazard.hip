// Probe: does a v_mfma issued right behind an EXEC that was just restored (from zero / from one lane) write all of its result?
// (profiles/probes_r04.md section 2: a Gram kernel whose wave-uniform `continue` was compiled as s_and_saveexec / s_cbranch_execz /
//  s_or exec around MFMAs lost a quarter-wave of one accumulator register now and then.)
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mfma_hazard exec_mfma_hazard.hip ; run: ./exec_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 g16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: no EXEC change (control); 1: exec = 0, s_cbranch_execz taken, restore; 2: exec = 0, restore (no branch); 3: exec = 1 (lane 0), a
// VALU move under it, restore.  GAP: s_nop wait states between the restore and the MFMA.
template <int MODE, int GAP>
__global__ void __launch_bounds__(512, 1) probe(int iters, unsigned* bad, unsigned* where) {
    const int lane = threadIdx.x & 63;
    g16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(1 + ((lane + i) & 3));
        b[i] = (_Float16)(1 + ((lane * 3 + i) & 1));
    }
    // reference result of three MFMAs, computed with EXEC untouched
    f32x16 want;
    for (int r = 0; r < 16; ++r) want[r] = 0.f;
    for (int k = 0; k < 3; ++k) want = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, want, 0, 0, 0);
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x16 acc0, acc1;
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        asm volatile("" : "+v"(acc0), "+v"(acc1));
        unsigned long long sv;
        unsigned dummy = 0;
        if (MODE == 0)
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "s_nop 15\ns_nop 15\n"
                : [c0] "+v"(acc0), [c1] "+v"(acc1), [sv] "=s"(sv), [d] "+v"(dummy)
                : [a] "v"(a), [b] "v"(b));
        else
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "v_mfma_f32_32x32x16_f16 %[c0], %[a], %[b], %[c0]\n"
                "s_mov_b64 %[sv], exec\n"
                "s_mov_b64 exec, %[m]\n"
                ".if %[br]\n"
                "s_cbranch_execz 1f\n"
                "v_mov_b32 %[d], 1\n"
                "1:\n"
                ".endif\n"
                ".if %[mv]\n"
                "v_mov_b32 %[d], 1\n"
                ".endif\n"
                "s_mov_b64 exec, %[sv]\n"
                ".if %[gap]\n"
                "s_nop %[gapm1]\n"
                ".endif\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "v_mfma_f32_32x32x16_f16 %[c1], %[a], %[b], %[c1]\n"
                "s_nop 15\ns_nop 15\n"
                : [c0] "+v"(acc0), [c1] "+v"(acc1), [sv] "=&s"(sv), [d] "+v"(dummy)
                : [a] "v"(a), [b] "v"(b), [m] "n"(MODE == 3 ? 1 : 0), [br] "n"(MODE == 1), [mv] "n"(MODE == 3), [gap] "n"(GAP > 0),
                  [gapm1] "n"(GAP > 0 ? GAP - 1 : 0));
        for (int r = 0; r < 16; ++r) {
            if (acc1[r] != want[r]) {
                ++nbad;
                where[(r * 64 + lane) & 1023] = 1 + (unsigned)__builtin_bit_cast(unsigned, acc1[r]);
            }
            if (acc0[r] != want[r]) nbad += 1u << 16;
        }
    }
    if (nbad) atomicAdd(bad, nbad & 0xffffu), atomicAdd(bad + 1, nbad >> 16);
}

template <int MODE, int GAP>
static void run(const char* what, int iters) {
    unsigned *bad, *where;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&where, 4096);
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(where, 0, 4096);
    hipLaunchKernelGGL((probe<MODE, GAP>), dim3(2048), dim3(512), 0, 0, iters, bad, where);
    (void)hipDeviceSynchronize();
    unsigned h[2], w[1024];
    (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(w, where, 4096, hipMemcpyDeviceToHost);
    int nregs = 0, first = -1;
    for (int i = 0; i < 1024; ++i)
        if (w[i]) {
            ++nregs;
            if (first < 0) first = i;
        }
    printf("%-58s wrong (register, lane) results behind the restore: %u of %.3g; in the MFMAs ahead of it: %u", what, h[0],
           2048.0 * 512 * iters * 16, h[1]);
    if (first >= 0) printf("   [%d distinct (r, lane), first r=%d lane=%d value %g]", nregs, first / 64, first % 64, __builtin_bit_cast(float, w[first] - 1));
    printf("\n");
    (void)hipFree(bad);
    (void)hipFree(where);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    run<0, 0>("control: EXEC untouched", iters);
    run<1, 0>("exec = 0, s_cbranch_execz taken, restore, MFMA", iters);
    run<1, 1>("exec = 0, branch, restore, s_nop 0, MFMA", iters);
    run<1, 4>("exec = 0, branch, restore, s_nop 3, MFMA", iters);
    run<2, 0>("exec = 0 (no branch), restore, MFMA", iters);
    run<2, 4>("exec = 0 (no branch), restore, s_nop 3, MFMA", iters);
    run<3, 0>("exec = lane 0, v_mov, restore, MFMA", iters);
    run<3, 4>("exec = lane 0, v_mov, restore, s_nop 3, MFMA", iters);
    return 0;
}
