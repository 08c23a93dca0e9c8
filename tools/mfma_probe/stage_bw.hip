// What the memory side of a 3x3 convolution's staging and epilogue is worth on its own (no MFMA, no LDS): 256 persistent workgroups of
// 512 threads walk 16 x 32 pixel tiles of a C-channel H x W fp32 map in 32-channel chunks, exactly conv_x3p's item order.
//   load dw : the kernel's pattern - a thread takes patch position tid (18 x 34 positions) of each of the four octets: 8 dword loads per
//             item (one per channel plane), 40 per thread and chunk;
//   load x4 : a thread takes (octet, row, 4-pixel group) of the 18 x 32 aligned interior: 8 dwordx4 loads; the two halo columns as dwords;
//   store dw: the kernel's epilogue - per lane 64 dword stores (lane = (octet, pixel): 4 channels x 4 pixel groups x 4 channel groups);
//   store x4: the same values as 16 dwordx4 stores (4 consecutive pixels of one channel per lane).
// hipcc --offload-arch=gfx950 -O3 -o stage_bw stage_bw.hip ; ./stage_bw [C] [H]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

template <int MODE, int ORDER = 0>
__global__ void __launch_bounds__(512, 2) load_kernel(const float* __restrict__ x, float* __restrict__ sink, int C, int H, int W, int tiles_x, int tiles) {
    const int tid = threadIdx.x;
    const int plane = H * W;
    const int gq = (int)(blockIdx.x & 7) * (gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    // ORDER 0: a contiguous run of tiles per workgroup; ORDER 1: XCD k owns the k-th eighth of the tiles, its workgroups take them in turn
    // (the workgroups of an XCD are on horizontally adjacent tiles at any time)
    const int per_xcd = tiles / 8, wg_x = gridDim.x >> 3;
    const int t0 = ORDER == 0 ? (int)((long)gq * tiles / gridDim.x) : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int t1 = ORDER == 0 ? (int)((long)(gq + 1) * tiles / gridDim.x) : (int)((blockIdx.x & 7) + 1) * per_xcd;
    const int dt = ORDER == 0 ? 1 : wg_x;
    float acc = 0.f;
    for (int t = t0; t < t1; t += dt) {
        const int x0 = (t % tiles_x) * 32, y0 = (t / tiles_x) * 16;
        for (int ch = 0; ch < C / 32; ++ch) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long)ch * 32 * plane), 0, (unsigned)plane * 128u, 0x00020000);
            if (MODE == 0) {
                // positions tid (4 octets) and 512 + tid % 100 of octet tid / 100
                const int ra = tid / 34, ca = tid % 34;
                const int iy = y0 + ra - 1, ix = x0 + ca - 1;
                const unsigned va = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (unsigned)(iy * W + ix) * 4u : OOB;
                const int o4 = tid / 100, pb = 512 + tid % 100, rb = pb / 34, cb = pb % 34;
                const int iyb = y0 + rb - 1, ixb = x0 + cb - 1;
                const unsigned vb = (tid < 400 && (unsigned)iyb < (unsigned)H && (unsigned)ixb < (unsigned)W) ? (unsigned)(o4 * 8 * plane + iyb * W + ixb) * 4u : OOB;
                float v[5][8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        v[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, k < 4 ? va : vb, ((k < 4 ? 8 * k : 0) + c) * plane * 4, 0));
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc += v[k][c];
            } else {
                // groups: (octet 4, row 18, group 8) = 576: thread tid -> group tid; threads 0-63 also group 512 + tid; halo: 144 (octet, row, side) items of 8 dwords: threads 64-207
                f4 v[8];
                float hv[8];
                {
                    const int o = tid / 144, rem = tid % 144, r = rem / 8, g = rem % 8;
                    const int iy = y0 + r - 1, ix = x0 + 4 * g;
                    const unsigned va = ((unsigned)iy < (unsigned)H && ix < W) ? (unsigned)(o * 8 * plane + iy * W + ix) * 4u : OOB;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, va, c * plane * 4, 0));
                }
                const bool second = tid < 64;
                f4 w[8];
                {
                    const int idx = 512 + tid;
                    const int o = idx / 144, rem = idx % 144, r = rem / 8, g = rem % 8;
                    const int iy = y0 + r - 1, ix = x0 + 4 * g;
                    const unsigned va = (second && (unsigned)iy < (unsigned)H && ix < W) ? (unsigned)(o * 8 * plane + iy * W + ix) * 4u : OOB;
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, va, c * plane * 4, 0));
                }
                {
                    const int h = tid - 64;
                    const int o = h / 36, rem = h % 36, r = rem / 2, side = rem % 2;
                    const int iy = y0 + r - 1, ix = x0 - 1 + 33 * side;
                    const unsigned va = (h >= 0 && h < 144 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (unsigned)(o * 8 * plane + iy * W + ix) * 4u : OOB;
#pragma unroll
                    for (int c = 0; c < 8; ++c) hv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, va, c * plane * 4, 0));
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) acc += v[c][0] + v[c][1] + v[c][2] + v[c][3] + w[c][0] + w[c][1] + w[c][2] + w[c][3] + hv[c];
            }
        }
    }
    if (acc == 12345.678f) sink[tid] = acc;
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) store_kernel(float* __restrict__ y, int C, int H, int W, int tiles_x, int tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, oct = lane >> 4;
    const int plane = H * W;
    const int gq = (int)(blockIdx.x & 7) * (gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int ncot = C / 64;
    const int items = tiles * ncot;
    const int t0 = (int)((long)gq * items / gridDim.x), t1 = (int)((long)(gq + 1) * items / gridDim.x);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, (unsigned)C * (unsigned)plane * 4u, 0x00020000);
    for (int it = t0; it < t1; ++it) {
        const int t = it % tiles, cot = it / tiles;
        const int x0 = (t % tiles_x) * 32, y0 = (t / tiles_x) * 16;
        const unsigned so = (unsigned)((cot * 64) * plane + y0 * W + x0) * 4u;
        const float val = (float)it;
        if (MODE == 0) {
            // lane: channel 16 i + 4 oct + r, pixel (row 2 wave + (g >> 1), column 16 (g & 1) + px)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned v = (unsigned)((4 * oct) * plane + (2 * wave) * W + 16 * (g & 1) + px) * 4u;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val + r), rs, v, so + (unsigned)((16 * i + r) * plane + (g >> 1) * W) * 4u, 0);
                }
        } else {
            // lane: channel 16 i + 4 oct + (px & 3), pixels 4 (px >> 2) ... + 3 of column half g & 1, row 2 wave + (g >> 1)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned v = (unsigned)((4 * oct + (px & 3)) * plane + (2 * wave) * W + 16 * (g & 1) + 4 * (px >> 2)) * 4u;
                    const f4 q = {val, val + 1, val + 2, val + 3};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rs, v, so + (unsigned)((16 * i) * plane + (g >> 1) * W) * 4u, 0);
                }
        }
    }
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 1024, W = H;
    const int tiles_x = (W + 31) / 32, tiles = tiles_x * ((H + 15) / 16);
    float *x, *y, *sink;
    const size_t bytes = (size_t)C * H * W * 4;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(x, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double mb) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s C %d @ %d: %8.1f us  %7.2f TB/s of useful bytes\n", name, C, H, ms * 100.0, mb / (ms * 100.0));
    };
    const double mb_in = (double)bytes / 1e6;
    for (int g : {256, 512}) {
        printf("grid %d\n", g);
        timeit("load dw", [&] { hipLaunchKernelGGL(load_kernel<0>, dim3(g), dim3(512), 0, 0, x, sink, C, H, W, tiles_x, tiles); }, mb_in);
        timeit("load dw il", [&] { hipLaunchKernelGGL((load_kernel<0, 1>), dim3(g), dim3(512), 0, 0, x, sink, C, H, W, tiles_x, tiles); }, mb_in);
        timeit("load x4 il", [&] { hipLaunchKernelGGL((load_kernel<1, 1>), dim3(g), dim3(512), 0, 0, x, sink, C, H, W, tiles_x, tiles); }, mb_in);
        timeit("load x4", [&] { hipLaunchKernelGGL(load_kernel<1>, dim3(g), dim3(512), 0, 0, x, sink, C, H, W, tiles_x, tiles); }, mb_in);
        timeit("store dw", [&] { hipLaunchKernelGGL(store_kernel<0>, dim3(g), dim3(512), 0, 0, y, C, H, W, tiles_x, tiles); }, mb_in);
        timeit("store x4", [&] { hipLaunchKernelGGL(store_kernel<1>, dim3(g), dim3(512), 0, 0, y, C, H, W, tiles_x, tiles); }, mb_in);
    }
    CK(hipGetLastError());
    return 0;
}
