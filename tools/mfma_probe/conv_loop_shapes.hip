// Upper bounds for the K loop of the fp16x3 3x3 convolution in four structures, all operands re-read from LDS by ds_read_b128 on
// random fp16 data, no staging and no barriers (what is left when every other cost of the kernel is hidden):
//   A  conv_x3w as shipped : 2 workgroups / CU (2 waves / SIMD, 256 registers), wave tile 64 couts x 64 px as 2 x 2 accumulators of
//                            v_mfma_f32_32x32x16_f16, 16-channel chunks: per tap 4 + 4 fragment reads for 12 MFMAs
//   B  same occupancy, v_mfma_f32_16x16x32_f16, 32-channel chunks: wave tile 64 x 64 as 4 x 4 accumulators, per tap 8 + 8 reads for 48 MFMAs
//   C  1 workgroup / CU (1 wave / SIMD, 512 registers), 16x16x32, wave tile 64 couts x 128 px as 4 x 8: per tap 8 + 16 reads for 96 MFMAs
//   D  structure C on 32x32x16 (wave tile 64 x 128 as 2 x 4, 16-channel chunks: per tap 4 + 8 reads for 24 MFMAs)
// Every structure folds its sums into fp32 masters once per chunk, like the kernel.  Prints us per launch, algorithmic TFLOP/s
// (3 MFMAs per product block), MFMA-pipe occupancy in cycles and the in-kernel clock.   hipcc --offload-arch=gfx950 -O3 -o conv_loop_shapes conv_loop_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

// A: 2 x 2 accumulators 32x32, K = 16 per tap
__global__ void __launch_bounds__(256, 2) loopA(const _Float16* src, float* out, unsigned long long* st, int chunks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[58 * 1024];
    for (int i = threadIdx.x; i < 58 * 1024 / 2; i += 256) reinterpret_cast<_Float16*>(smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    const unsigned char* Pl = smem;              // [part][octet][344 pos][16 B]
    const unsigned char* Wl = smem + 4 * 5504;   // [tap][part][octet][64 co][16 B]
    const int b_base = half * 5504 + ((2 * wave) * 34 + j) * 16, a_base = half * 1024 + j * 16;
    f32x16 acc[2][2] = {}, master[2][2] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    typedef f16x8 AFrag[2];
    typedef f16x8 BFrag[2][2];
    auto load_a = [&](AFrag& a, int tap, int t) {
#pragma unroll
        for (int part = 0; part < 2; ++part) a[part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 2048 + t * 512);
    };
    auto load_b = [&](BFrag& b, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int part = 0; part < 2; ++part) b[row][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 2 * 5504 + ((row + ky) * 34 + kx) * 16);
    };
    auto mfma_half = [&](const AFrag& a, const BFrag& b, int t) {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[row][0], acc[t][row], 0, 0, 0);
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[row][1], acc[t][row], 0, 0, 0);
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[row][0], acc[t][row], 0, 0, 0);
        }
    };
    AFrag a0, a1;
    BFrag bx, by;
#define XW_TAP(TAP, BCUR, BNXT, HAS_NEXT) \
    do {                                  \
        load_a(a1, TAP, 1);               \
        if (HAS_NEXT) load_b(BNXT, (TAP) + 1); \
        FENCE();                          \
        mfma_half(a0, BCUR, 0);           \
        FENCE();                          \
        if (HAS_NEXT) load_a(a0, (TAP) + 1, 0); \
        FENCE();                          \
        mfma_half(a1, BCUR, 1);           \
        FENCE();                          \
    } while (0)
    for (int ch = 0; ch < chunks; ++ch) {
        load_b(bx, 0);
        load_a(a0, 0, 0);
        FENCE();
        XW_TAP(0, bx, by, true); XW_TAP(1, by, bx, true); XW_TAP(2, bx, by, true); XW_TAP(3, by, bx, true); XW_TAP(4, bx, by, true);
        XW_TAP(5, by, bx, true); XW_TAP(6, bx, by, true); XW_TAP(7, by, bx, true); XW_TAP(8, bx, by, false);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[t][row][r] = fmaf(acc[t][row][r], 0.5f, master[t][row][r]);
                    acc[t][row][r] = 0.f;
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int row = 0; row < 2; ++row) for (int r = 0; r < 16; ++r) s += master[t][row][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

// B / C: NA x NB accumulators 16x16, K = 32 per tap; patch [part][4 octets][PPOS][16 B], filters [tap][part][4 octets][64 co][16 B]
template <int NB, int OCC, int PPOS, int ROWS_PER_WAVE, int ORD>
__global__ void __launch_bounds__(256, OCC) loopBC(const _Float16* src, float* out, unsigned long long* st, int chunks) {
    constexpr int PLANE = PPOS * 16, PATCH = 8 * PLANE, FIL = 9 * 8192;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < (PATCH + FIL) / 2; i += 256) reinterpret_cast<_Float16*>(smem)[i] = src[i & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane & 15, oct = lane >> 4;
    const unsigned char* Pl = smem;
    const unsigned char* Wl = smem + PATCH;
    const int b_base = oct * PLANE + ((ROWS_PER_WAVE * wave) * 34 + px) * 16, a_base = oct * 1024 + px * 16;
    f32x4 acc[4][NB] = {}, master[4][NB] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 b[NB][2], a[2][2];
    auto load_bg = [&](int g, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int part = 0; part < 2; ++part)
            b[g][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 4 * PLANE + (((g >> 1) + ky) * 34 + (g & 1) * 16 + kx) * 16);
    };
    auto load_ai = [&](int buf, int tap, int i) {
#pragma unroll
        for (int part = 0; part < 2; ++part) a[buf][part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 4096 + i * 256);
    };
#pragma unroll
    for (int g = 0; g < NB; ++g) load_bg(g, 0);
    load_ai(0, 0, 0);
    FENCE();
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cur = i & 1;
                load_ai(cur ^ 1, i == 3 ? (tap + 1) % 9 : tap, (i + 1) & 3);
                FENCE();
                if constexpr (ORD == 0) {
#pragma unroll
                    for (int g = 0; g < NB; ++g) {
                        acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][1], b[g][0], acc[i][g], 0, 0, 0);
                        acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[g][1], acc[i][g], 0, 0, 0);
                        acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[g][0], acc[i][g], 0, 0, 0);
                        if (i == 3) {
                            FENCE();
                            load_bg(g, (tap + 1) % 9);
                            FENCE();
                        }
                    }
                } else {  // no two consecutive MFMAs on one accumulator
#pragma unroll
                    for (int g = 0; g < NB; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][1], b[g][0], acc[i][g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NB; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[g][1], acc[i][g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NB; ++g) {
                        acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[g][0], acc[i][g], 0, 0, 0);
                        if (i == 3) {
                            FENCE();
                            load_bg(g, (tap + 1) % 9);
                            FENCE();
                        }
                    }
                }
                FENCE();
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    master[i][g][r] = fmaf(acc[i][g][r], 0.5f, master[i][g][r]);
                    acc[i][g][r] = 0.f;
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int g = 0; g < NB; ++g) for (int r = 0; r < 4; ++r) s += master[i][g][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

// D: 1 workgroup / CU, 32x32x16, wave tile 64 couts x 128 px = 2 x 4 accumulators, 16-channel chunks
__global__ void __launch_bounds__(256, 1) loopD(const _Float16* src, float* out, unsigned long long* st, int chunks) {
    constexpr int PLANE = 624 * 16, PATCH = 4 * PLANE, FIL = 9 * 4096;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < (PATCH + FIL) / 2; i += 256) reinterpret_cast<_Float16*>(smem)[i] = src[i & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    const unsigned char* Pl = smem;
    const unsigned char* Wl = smem + PATCH;
    const int b_base = half * PLANE + ((4 * wave) * 34 + j) * 16, a_base = half * 1024 + j * 16;
    f32x16 acc[2][4] = {}, master[2][4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 b[4][2], a[2][2];
    auto load_bg = [&](int row, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int part = 0; part < 2; ++part) b[row][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 2 * PLANE + ((row + ky) * 34 + kx) * 16);
    };
    auto load_ai = [&](int buf, int tap, int t) {
#pragma unroll
        for (int part = 0; part < 2; ++part) a[buf][part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 2048 + t * 512);
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) load_bg(g, 0);
    load_ai(0, 0, 0);
    FENCE();
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                load_ai(t ^ 1, t == 1 ? (tap + 1) % 9 : tap, t ^ 1);
                FENCE();
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[row][0], acc[t][row], 0, 0, 0);
                    acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[row][1], acc[t][row], 0, 0, 0);
                    acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[row][0], acc[t][row], 0, 0, 0);
                    if (t == 1) {
                        FENCE();
                        load_bg(row, (tap + 1) % 9);
                        FENCE();
                    }
                }
                FENCE();
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[t][row][r] = fmaf(acc[t][row][r], 0.5f, master[t][row][r]);
                    acc[t][row][r] = 0.f;
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int row = 0; row < 4; ++row) for (int r = 0; r < 16; ++r) s += master[t][row][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Variant {
    const char* name;
    void (*launch)(const _Float16*, float*, unsigned long long*, int, hipStream_t);
    int wgs, chunks;          // chunks of the variant's own size per workgroup
    double flop_per_wg;       // algorithmic: 2 x 9 x channels x 64 couts x pixels of the workgroup tile
    double mfma_cycles_simd;  // matrix-pipe cycles per SIMD and workgroup
    int wg_per_cu;
};

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    _Float16* src; float* out; unsigned long long* st;
    hipMalloc(&src, 65536 * 2 + 64 * 1024); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&st, 1024 * 8);
    std::vector<_Float16> h(65536 + 32 * 1024);
    srand(1);
    for (auto& v : h) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 4000.f);  // full mantissas, magnitudes like the scaled chunks
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)loopBC<4, 2, 352, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 352 * 16 + 9 * 8192);
    hipFuncSetAttribute((const void*)loopBC<4, 2, 352, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 352 * 16 + 9 * 8192);
    hipFuncSetAttribute((const void*)loopBC<8, 1, 624, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 624 * 16 + 9 * 8192);
    hipFuncSetAttribute((const void*)loopBC<8, 1, 624, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 624 * 16 + 9 * 8192);
    hipFuncSetAttribute((const void*)loopD, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 624 * 16 + 9 * 4096);
    // equal work per CU: 512 channels of a 64-cout x 512-px region
    const int CH = 512;
    Variant v[6] = {
        {"A  32x32x16 2wg/cu 64x64 ", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL(loopA, dim3(512), dim3(256), 0, q, s, o, t, c); },
         512, CH / 16, 2.0 * 9 * CH * 64 * 256, (double)(CH / 16) * 9 * 12 * 32, 2},
        {"B0 16x16x32 2wg/cu 64x64 ", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL((loopBC<4, 2, 352, 2, 0>), dim3(512), dim3(256), 8 * 352 * 16 + 9 * 8192, q, s, o, t, c); },
         512, CH / 32, 2.0 * 9 * CH * 64 * 256, (double)(CH / 32) * 9 * 48 * 16, 2},
        {"C0 16x16x32 1wg/cu 64x128", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL((loopBC<8, 1, 624, 4, 0>), dim3(256), dim3(256), 8 * 624 * 16 + 9 * 8192, q, s, o, t, c); },
         256, CH / 32, 2.0 * 9 * CH * 64 * 512, (double)(CH / 32) * 9 * 96 * 16, 1},
        {"B1 16x16x32 2wg/cu 64x64 ", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL((loopBC<4, 2, 352, 2, 1>), dim3(512), dim3(256), 8 * 352 * 16 + 9 * 8192, q, s, o, t, c); },
         512, CH / 32, 2.0 * 9 * CH * 64 * 256, (double)(CH / 32) * 9 * 48 * 16, 2},
        {"C1 16x16x32 1wg/cu 64x128", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL((loopBC<8, 1, 624, 4, 1>), dim3(256), dim3(256), 8 * 624 * 16 + 9 * 8192, q, s, o, t, c); },
         256, CH / 32, 2.0 * 9 * CH * 64 * 512, (double)(CH / 32) * 9 * 96 * 16, 1},
        {"D  32x32x16 1wg/cu 64x128", [](const _Float16* s, float* o, unsigned long long* t, int c, hipStream_t q) { hipLaunchKernelGGL(loopD, dim3(256), dim3(256), 4 * 624 * 16 + 9 * 4096, q, s, o, t, c); },
         256, CH / 16, 2.0 * 9 * CH * 64 * 512, (double)(CH / 16) * 9 * 24 * 32, 1},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> us[6], clk[6], cyc[6];
    const int reps = 20;
    for (int r = 0; r < rounds + 1; ++r)
        for (int k = 0; k < 6; ++k) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < reps; ++i) v[k].launch(src, out, st, v[k].chunks, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            if (hipGetLastError() != hipSuccess) { printf("launch failed: %s\n", v[k].name); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hs(2 * v[k].wgs);
            hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> c, f;
            for (int i = 0; i < v[k].wgs; ++i) { c.push_back((double)hs[2 * i]); f.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 0.1); }
            std::sort(c.begin(), c.end()); std::sort(f.begin(), f.end());
            if (r == 0) continue;  // warm-up round
            us[k].push_back(ms * 1e3 / reps); cyc[k].push_back(c[c.size() / 2]); clk[k].push_back(f[f.size() / 2]);
        }
    for (int k = 0; k < 6; ++k) {
        std::sort(us[k].begin(), us[k].end()); std::sort(cyc[k].begin(), cyc[k].end()); std::sort(clk[k].begin(), clk[k].end());
        const double m = us[k][us[k].size() / 2], cy = cyc[k][cyc[k].size() / 2], ck = clk[k][clk[k].size() / 2];
        printf("%s  %8.1f us (min %8.1f)  %7.1f TF algorithmic  loop cycles %9.0f  pipe occupancy %.3f  clock %.3f GHz\n", v[k].name, m, us[k][0],
               v[k].flop_per_wg * v[k].wgs / m / 1e6, cy, v[k].mfma_cycles_simd * v[k].wg_per_cu / cy, ck);
    }
    return 0;
}
