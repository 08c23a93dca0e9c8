// 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp16x3 split: 2.25x fewer matrix products than conv_x3w.hip.
//
// conv_x3w.hip is bound by the clock the chip holds under its MFMA load (profiles/probes_r02.md): stalls, LDS traffic and issue
// slots saved there come back as a lower clock, only removed matrix work pays.  Winograd's minimal filtering removes it:
//   Y (2x2 outputs of a tile) = A^T [ sum over channels of (G g G^T) .* (B^T d B) ] A
// with d the 4x4 input window of the tile: 16 element-wise products per 4 outputs instead of 36.  The sum over channels of each
// of the 16 "positions" is an ordinary GEMM  M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile]  and runs on
// v_mfma_f32_32x32x16_f16 in the fp16x3 arithmetic of conv_x3.hip: U (filters, transformed in fp64 once per weight version) and
// V (transformed in fp32 while staging, +-1 additions only) are each split into two fp16 parts, three products per block.
// Error against fp64 per layer: 1.1-1.3x the fp32 CPU convolution's (tools/probe_winograd_numerics.py; tests/test_conv_wino_gpu.py).
//
// Structure (gfx950):
//   * workgroup = 8 waves = 64 output channels x (8 rows x 32 px) = 64 tiles of 2x2, ALL 16 positions: 65,536 accumulators =
//     128 registers per lane, ONE workgroup per CU (two waves per SIMD, 256 registers each).  Wave w owns positions 2w, 2w+1
//     for the whole 64 x 64 block: per position 4 filter + 4 input fragments for 12 MFMAs.
//   * the 128 registers ARE the fp32 master sums: a (position, block) pair gets only three MFMAs per chunk, so its chunk sum is a
//     short-lived 16-register value that starts from zero and is folded - one plain FMA per value, round to nearest, times the
//     chunk's inverse power-of-two scale - into the masters right away.  Per-chunk scales like conv_x3w's.
//   * K chunk = 16 input channels.  Filters never touch LDS: the bank holds every (chunk, position, 32-channel block, part)
//     fragment in MFMA lane order, a wave's A operand is ONE global_load_dwordx4 (1 KiB contiguous, L2-resident: the launch
//     maps output-channel tiles to XCDs), requested one position ahead.
//   * inputs: raw fp32 patch [16 ch][10 rows][36] in LDS (single buffer), transformed by all 512 lanes - lane = (tile,
//     channel pair): 4x4 window -> B^T d B (32 additions per channel) -> scale, split, ds_write_b32 of the (hi, hi) and (lo, lo)
//     pairs into V [position][part][octet][tile][8 ch], double-buffered (2 x 64 KiB): the transform of chunk c + 1 and the
//     products of chunk c overlap - waves 0-3 transform first, waves 4-7 multiply first, so the two waves of a SIMD are in
//     complementary phases.
//   * epilogue: the 16 positions of an output live in 8 different waves: they meet in LDS (two halves of 32 channels, padded
//     [position][tile][36]), every lane then owns 4 channels x one tile: A^T M A, un-scale, bias / ReLU / mask / accumulate.
#include <stdlib.h>

#include <algorithm>

#include "../../maua-style_amd/csrc/common.hpp"
#include "maua_wino.h"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int WN_COT = 64;
constexpr int WN_ROWS = 8, WN_COLS = 32;
constexpr int WN_PR = 10;                             // raw patch: 10 rows of 34 columns, one row = 64 floats of LDS
constexpr int WN_RAW_BYTES = 16 * WN_PR * 256;        // one chunk of the raw patch: [16 ch][10 rows][64 floats] = 40,960
constexpr int WN_E_STRIDE = 36;                       // epilogue exchange: [pos][tile][36 floats] (32 channels + pad)
constexpr int WN_EXCH_BYTES = 16 * 64 * WN_E_STRIDE * 4;                                         // 147,456
constexpr int WN_SMEM_BYTES = WN_EXCH_BYTES > 3 * WN_RAW_BYTES ? WN_EXCH_BYTES : 3 * WN_RAW_BYTES;  // ring of three raw chunks
constexpr int WN_BANK_HDR = 4096;                     // bank header: [0] max |U| bits (packing scratch), [1] 1 / scale
constexpr int WN_U_BYTES = 16 * 4 * 1024;             // per (chunk, cout tile): [pos][32-ch block][part][lane][16 B] = 65,536

__device__ __forceinline__ unsigned wn_cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float wn_f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float wn_f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// ---------------------------------------------------------------------------------------------------------
// filter bank: U = G g G^T in fp64, scaled by a power of two (max |U| into [32, 64)), two fp16 parts, MFMA lane order
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wn_g3(int k, double a, double b, double c) {  // row k of G times (a, b, c)
    return k == 0 ? a : k == 1 ? 0.5 * (a + b + c) : k == 2 ? 0.5 * (a - b + c) : c;
}
// U[xi][nu] of the 3x3 filter of (output o, input i) in the direction's roles
__device__ __forceinline__ void wn_filter_u(const float* __restrict__ w, int cin_w, int o, int i, int backward, double (&U)[16]) {
    double g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        g[t / 3][t % 3] = backward ? (double)w[((int64_t)i * cin_w + o) * 9 + (8 - t)] : (double)w[((int64_t)o * cin_w + i) * 9 + t];
    double h[4][3];  // G g
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int b = 0; b < 3; ++b) h[xi][b] = wn_g3(xi, g[0][b], g[1][b], g[2][b]);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) U[xi * 4 + nu] = wn_g3(nu, h[xi][0], h[xi][1], h[xi][2]);
}

__global__ void wino_umax_kernel(const float* __restrict__ w, int cout, int cin, int backward, unsigned* __restrict__ hdr) {
    const int CO = backward ? cin : cout, CI = backward ? cout : cin;
    const int64_t total = (int64_t)CO * CI;
    float m = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        double U[16];
        wn_filter_u(w, cin, (int)(e / CI), (int)(e % CI), backward, U);
#pragma unroll
        for (int p = 0; p < 16; ++p) m = fmaxf(m, fabsf((float)U[p]));
    }
    m = wave_max_nonneg(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(hdr, __builtin_bit_cast(unsigned, m));  // (non-negative floats order like their bits)
}

__global__ void wino_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ bank, int cout, int cin, int backward) {
    const int CO = backward ? cin : cout, CI = backward ? cout : cin;
    const int nchunk = (CI + 15) / 16, ntile = (CO + WN_COT - 1) / WN_COT;
    const float m = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(bank)[0]);
    float scale = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
        scale = __builtin_bit_cast(float, (unsigned)(127 + 5 - max(e, -100)) << 23);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(bank)[1] = 1.f / scale;
    unsigned char* body = bank + WN_BANK_HDR;
    const int64_t total = (int64_t)nchunk * ntile * 2 * 64;  // (chunk, tile, 32-channel block, lane)
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int lane = (int)(r % 64);
        r /= 64;
        const int cb = (int)(r % 2);
        r /= 2;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * WN_COT + cb * 32 + (lane & 31);
        const int i0 = chunk * 16 + (lane >> 5) * 8;
        _Float16 hi[16][8], lo[16][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double U[16];
            const bool ok = o < CO && i0 + c < CI;
            if (ok) wn_filter_u(w, cin, o, i0 + c, backward, U);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float v = ok ? (float)(U[p] * (double)scale) : 0.f;
                const _Float16 h = (_Float16)v;
                hi[p][c] = h;
                lo[p][c] = (_Float16)(v - (float)h);
            }
        }
        unsigned char* dst = body + (((int64_t)chunk * ntile + tile) * 16) * 4096 + cb * 2048 + lane * 16;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f16x8 H, L;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                H[c] = hi[p][c];
                L[c] = lo[p][c];
            }
            *reinterpret_cast<f16x8*>(dst + p * 4096) = H;
            *reinterpret_cast<f16x8*>(dst + p * 4096 + 1024) = L;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// the convolution
// ---------------------------------------------------------------------------------------------------------
#ifdef WN_STAMP
#define WN_MARK(k)                                                                                                 \
    do {                                                                                                           \
        if (lane == 0 && p.mask) {                                                                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                            \
            const_cast<float*>(p.mask)[(((int64_t)blockIdx.x * 8 + wave) * 64 + min(ch, 62)) * 8 + (k)] = __builtin_bit_cast(float, (unsigned)t_); \
        }                                                                                                          \
    } while (0)
#else
#define WN_MARK(k) ((void)0)
#endif

// One LDS-DMA instruction: 64 lanes x 4 bytes from a buffer resource (out-of-range lanes deliver 0) to 256 consecutive LDS bytes.
__device__ __forceinline__ void wn_dma_row(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <bool ACC, bool OM>
__global__ void __launch_bounds__(512, 2) conv_wino_kernel(ConvArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WN_SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, half = lane >> 5;
    const int n = blockIdx.z;
    const int ntile = (p.Cout + WN_COT - 1) / WN_COT;
    const int tiles_y = (p.OH + WN_ROWS - 1) / WN_ROWS;
    const int tiles_sp = p.tiles_x * tiles_y;
    // Output-channel tiles are pinned to XCDs (workgroup b runs on XCD b % 8): an XCD then streams only its own slices of the
    // filter bank (2 MiB per 64 channels at Cin = 512: L2-resident) and walks the spatial tiles in order (halo rows hit L2).
    int cotile, tile;
    {
        const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
        if ((ntile & 7) == 0) {
            cotile = xcd + 8 * (r / tiles_sp);
            tile = r % tiles_sp;
        } else if (ntile == 1 || ntile == 2 || ntile == 4) {
            const int f = 8 / ntile;
            cotile = xcd % ntile;
            tile = r * f + xcd / ntile;
        } else {
            cotile = id % ntile;
            tile = id / ntile;
        }
        if (tile >= tiles_sp || cotile >= ntile) return;  // whole workgroup leaves
    }
    const int co0 = cotile * WN_COT;
    const int x0 = (tile % p.tiles_x) * WN_COLS, y0 = (tile / p.tiles_x) * WN_ROWS;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    const float u_inv = reinterpret_cast<const float*>(bank)[1];

    // ---- this wave's two Winograd positions: row xi = wave / 2 of B^T d B, columns nu = (0, 1) or (2, 3)
    //      t[c] = d[rA][c] + sigma d[rB][c]   (rows by xi; tau = overall sign of the row combination, carried by the scales)
    //      A lane reads (p, q) = t[0, 1] and r = t[2]  (nu pair 0)   or   (p, q) = t[2, 3] and r = t[1]  (nu pair 1); then
    //      slot 0 = p - r = V[xi][0] or V[xi][2],   slot 1 = q + gamma r = V[xi][1] or -V[xi][3]   (gamma = +1 / -1)
    const int xi = wave >> 1, nup = wave & 1;
    const int rA = xi == 0 ? 0 : 1, rB = xi == 3 ? 3 : 2;
    const float sigma = xi == 1 ? 1.f : -1.f, tau = xi == 2 ? -1.f : 1.f;
    const float gamma = nup ? -1.f : 1.f, tau1 = nup ? -tau : tau;
    const int pos0 = xi * 4 + 2 * nup, pos1 = pos0 + 1;  // bank / exchange positions of slot 0 / 1

    // ---- raw patch ring in LDS: [buffer][16 ch][10 rows][64 floats]; column c of patch row R sits at (c + 32 ((R >> 1) & 1)) % 64
    //      (rows two apart - the two tile rows of a 32-tile block - land 32 banks apart: the window reads are conflict-free).
    //      Wave w brings in channels 2w, 2w + 1: 20 rows = 20 LDS-DMA instructions per chunk, no registers, no LDS store path.
    unsigned dma_voff[2];  // lane -> source column offset for a row without / with the rotation (out of range -> 0x80000000)
#pragma unroll
    for (int rot = 0; rot < 2; ++rot) {
        const int c = (lane - 32 * rot) & 63;
        const int ix = x0 + c - p.pad;
        dma_voff[rot] = (c < 34 && ix >= 0 && ix < p.W) ? (unsigned)ix * 4u : 0x80000000u;
    }
    const unsigned range = (unsigned)in_plane * 64u;  // 16 planes
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma_chunk = [&](int ch, int slot) {
        asm volatile("" : "+s"(ch));
        // buffer resource over the chunk's 16 planes (V#: base, stride 0, num_records in bytes, raw dword format)
        const uint64_t base = (uint64_t)(size_t)(xin + (int64_t)ch * 16 * in_plane);
        const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) & 0xffffu;
        asm volatile("s_nop 4" ::: "memory");  // (SGPRs written by v_readfirstlane need five wait states before a VMEM instruction reads them)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int R = 0; R < WN_PR; ++R) {
                const int iy = y0 + R - p.pad;
                const bool row_ok = iy >= 0 && iy < p.H;
                // a patch row outside the image: a descriptor of zero records - every lane is out of range and delivers 0
                const u32x4 rs = {b_lo, b_hi, row_ok ? range : 0u, 0x00020000u};
                const unsigned soff = row_ok ? (unsigned)(((2 * wave + e) * in_plane + iy * p.W) * 4) : 0u;
                wn_dma_row(rs, dma_voff[(R >> 1) & 1], soff, lds_base + slot * WN_RAW_BYTES + ((2 * wave + e) * WN_PR + R) * 256);
            }
    };

    // ---- window reads: lane (tile column n = j of a 32-tile block, channel octet = half): tile row trl = j / 16, tile column tc = j % 16
    const int trl = j >> 4, tc = j & 15;
    int rd64[2], rd32[2];  // [row A / B]: byte offsets inside a raw buffer (tile block 0, channel 0 of the octet) of (p, q) and of r
#pragma unroll
    for (int ab = 0; ab < 2; ++ab) {
        const int r = ab ? rB : rA;
        const int rot = 32 * ((trl + (r >> 1)) & 1);
        const int row = ((half * 8) * WN_PR + 2 * trl + r) * 256;
        rd64[ab] = row + ((2 * tc + 2 * nup + rot) & 63) * 4;
        rd32[ab] = row + ((2 * tc + 2 - nup + rot) & 63) * 4;
    }

    // ---- filter fragments straight from the bank (MFMA lane order, L2-resident): slot i -> position pos_i
    typedef u32x4 AFrag[2][2];  // [32-channel block][part]
    const unsigned char* __restrict__ ubody = bank + WN_BANK_HDR + (int64_t)cotile * 16 * 4096 + lane * 16;
    const int64_t u_chunk_stride = (int64_t)ntile * 16 * 4096;
    AFrag a0, a1;
    auto load_a = [&](int ch) {
        const unsigned char* g0 = ubody + (int64_t)ch * u_chunk_stride + pos0 * 4096;
        const unsigned char* g1 = ubody + (int64_t)ch * u_chunk_stride + pos1 * 4096;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                a0[cb][part] = *reinterpret_cast<const u32x4*>(g0 + (cb * 2 + part) * 1024);
                a1[cb][part] = *reinterpret_cast<const u32x4*>(g1 + (cb * 2 + part) * 1024);
            }
    };

    // ---- running sums: acc[slot][32-channel block][32-tile block]
    f32x16 acc[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][cb][tb][r] = 0.f;

    // The input operand of one 32-tile block: B^T d B for this wave's two positions from the raw window (fp32, +-1 additions),
    // a power-of-two scale PER TILE (= per MFMA column: both lanes of a column agree on it through one lane swap) that brings the
    // larger of the two positions' maxima over the 16 channels into [2^13, 2^14), the two-part fp16 split.  Returns 1 / scale.
    typedef f16x8 BFrag[2][2];  // [slot][part]
    auto make_b = [&](BFrag& b, const unsigned char* __restrict__ Rb, int tb) {
        float V[2][8];
        int o64a = rd64[0], o64b = rd64[1], o32a = rd32[0], o32b = rd32[1];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int o = tb * 1024 + c * (WN_PR * 256);
            const f32x2 pqA = *reinterpret_cast<const f32x2*>(Rb + o64a + o), pqB = *reinterpret_cast<const f32x2*>(Rb + o64b + o);
            const float rrA = *reinterpret_cast<const float*>(Rb + o32a + o), rrB = *reinterpret_cast<const float*>(Rb + o32b + o);
            const float pp = fmaf(sigma, pqB[0], pqA[0]), qq = fmaf(sigma, pqB[1], pqA[1]), rr = fmaf(sigma, rrB, rrA);
            V[0][c] = pp - rr;
            V[1][c] = fmaf(gamma, rr, qq);
            if (c & 1) asm volatile("" : "+v"(V[0][c]), "+v"(V[1][c]), "+v"(o64a), "+v"(o64b), "+v"(o32a), "+v"(o32b));
        }
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c += 2) m = fmaxf(m, fmaxf(fmaxf(fabsf(V[0][c]), fabsf(V[0][c + 1])), fmaxf(fabsf(V[1][c]), fabsf(V[1][c + 1]))));
        {   // the other channel octet of the same tile column lives in lane ^ 32
            // v_permlane32_swap: lanes 32-63 of the first register <-> lanes 0-31 of the second; with the maximum in both, one
            // register then holds the lower octet's value in every lane and the other the upper octet's.  (Inline asm: hipcc folds
            // the builtin's two results into one and drops the maximum below.  s_nop 1: VALU write -> permlane read.)
            float ma = m, mb = m;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
            m = fmaxf(ma, mb);
        }
        unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;  // biased exponent; 0 for an all-zero (or denormal) column
        eb = eb < 27u ? 27u : (eb > 227u ? 227u : eb);
        const float s = __builtin_bit_cast(float, (unsigned)(127 + 13 + 127 - (int)eb) << 23);
        const float inv = __builtin_bit_cast(float, (unsigned)((int)eb - 13) << 23);  // (the signs ride on the scales only)
        // two-part fp16 split of V s: hi = f16(V s) straight from the FMA (v_fma_mixlo / mixhi: one rounding), lo = f16(V s - hi) with
        // the residual exact in fp32 (v_fma_mix_f32 reads the f16 half it subtracts) - 2.5 instructions per value
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float si = s * (i ? tau1 : tau);
            u32x4 H, L;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h2;
                float r0, r1;
                asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(h2) : "v"(V[i][2 * q]), "v"(V[i][2 * q + 1]), "v"(si));
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(V[i][2 * q]), "v"(si), "v"(h2));
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(V[i][2 * q + 1]), "v"(si), "v"(h2));
                H[q] = h2;
                L[q] = wn_cvt_pk_f16(r0, r1);
            }
            b[i][0] = __builtin_bit_cast(f16x8, H);
            b[i][1] = __builtin_bit_cast(f16x8, L);
        }
        return inv;
    };
    // Three MFMAs into a short-lived sum that starts from zero, folded into the running sums (plain fp32 FMAs, round to nearest,
    // times the column's inverse scale - a power of two) right away: the MFMA adder only ever chains three steps, the running
    // sums see ONE rounding per 16 channels (conv_x3w's master accumulators without a second persistent register set).
    auto mfma3 = [&](const AFrag& a, const f16x8 (&b)[2], int cb) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[cb][1]), b[0], zero, 0, 0, 0);  // smallest terms first
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[cb][0]), b[1], t, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[cb][0]), b[0], t, 0, 0, 0);
    };
    auto fold = [&](f32x16& dst, const f32x16& t, float inv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = fmaf(t[r], inv, dst[r]);
    };
#define WN_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#define WN_PINB(x, bb) asm volatile("" : "+v"(x), "+v"(bb[0][0]), "+v"(bb[0][1]), "+v"(bb[1][0]), "+v"(bb[1][1]))
    // the four (slot, channel block) pairs of one tile block, software-pipelined over two short-lived sums; the empty asm
    // statements pin the order (each "rewrites" what the next step reads)
    auto block = [&](BFrag& b, int tb, float inv) {
        f32x16 t0 = mfma3(a0, b[0], 0);
        WN_PINB(t0, b);
        fold(acc[0][0][tb], t0, inv);
        WN_PINB(acc[0][0][tb], b);
        t0 = mfma3(a0, b[0], 1);
        WN_PINB(t0, b);
        fold(acc[0][1][tb], t0, inv);
        WN_PINB(acc[0][1][tb], b);
        t0 = mfma3(a1, b[1], 0);
        WN_PINB(t0, b);
        fold(acc[1][0][tb], t0, inv);
        WN_PINB(acc[1][0][tb], b);
        t0 = mfma3(a1, b[1], 1);
        fold(acc[1][1][tb], t0, inv);
    };

    // ---- channel loop.  Raw patches run three chunks deep in the ring (DMA of chunk c + 2 issued at the end of iteration c, after
    //      the filter fragments of chunk c + 1: vector-memory results return in issue order, so the wait for the fragments at the
    //      top of the next iteration also covers the DMA of ITS chunk and leaves the newest 20 DMA instructions in flight).
    //      One barrier per chunk: raw(c) complete for every wave's DMA share; everybody is done reading raw(c - 1).
    const int nch = p.Cin / 16;
    dma_chunk(0, 0);
    if (nch > 1) dma_chunk(1, 1);
    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (once: the first two raw chunks of this wave's share are in LDS)
#ifdef WN_STAMP
    if (lane == 0 && p.mask) {  // which CU / SIMD this wave runs on, shader clock and 100 MHz clock at loop start: slot 63
        float* d_ = const_cast<float*>(p.mask) + (((int64_t)blockIdx.x * 8 + wave) * 64 + 63) * 8;
        d_[0] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4));
        d_[1] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[2] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif
    // Iteration c: barrier (raw(c) is complete - see below - and everybody is done with raw(c - 1)); first tile block; the DMA of
    // raw(c + 2) into the slot raw(c - 1) left; second tile block; the filter fragments of chunk c + 1 (plain loads: the compiler
    // waits for them before the next iteration's first MFMA - and vector-memory results return in issue order, so that wait also
    // completes the DMA issued before them: raw(c + 2) is in LDS one whole iteration before its barrier).
    for (int ch = 0; ch < nch; ++ch) {
        WN_MARK(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");  // (no window read of the compiler's may move above the barrier)
        WN_MARK(1);
        const unsigned char* Rb = smem + (ch % 3) * WN_RAW_BYTES;
        BFrag b;
        float inv = make_b(b, Rb, 0);
        WN_MARK(2);
        block(b, 0, inv);
        WN_PINB(acc[1][1][0], b);
        WN_MARK(3);
        if (ch + 2 < nch) dma_chunk(ch + 2, (ch + 2) % 3);
        WN_MARK(4);
        inv = make_b(b, Rb, 1);
        WN_MARK(5);
        block(b, 1, inv);
        WN_MARK(6);
        if (ch + 1 < nch) {
            WN_PIN2(acc[1][0][1], acc[1][1][1]);  // (every MFMA that reads the fragments has issued)
            load_a(ch + 1);
        }
        WN_MARK(7);
    }
#ifdef WN_STAMP
    if (lane == 0 && p.mask) {
        float* d_ = const_cast<float*>(p.mask) + (((int64_t)blockIdx.x * 8 + wave) * 64 + 63) * 8;
        d_[3] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[4] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif

    // ---- epilogue: positions meet in LDS, 32 channels at a time; then lane = (4 channels (wave), tile (lane))
    const float inv = u_inv;
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = OM ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
    float* El = reinterpret_cast<float*>(smem);
    const int tr = lane >> 4;
    const int oy = y0 + 2 * tr, ox = x0 + 2 * tc;
    const bool pair_ok = (p.OW & 1) == 0;  // float2 stores need even row starts
    __syncthreads();  // (the last chunk's windows have been read: the ring becomes the exchange area)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();  // the reads of the first half are done
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][h][tb][4 * q], acc[i][h][tb][4 * q + 1], acc[i][h][tb][4 * q + 2], acc[i][h][tb][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(El + ((i ? pos1 : pos0) * 64 + tb * 32 + j) * WN_E_STRIDE + 8 * q + 4 * half) = v;
                }
        __syncthreads();
        f32x4 m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) m[q] = *reinterpret_cast<const f32x4*>(El + (q * 64 + lane) * WN_E_STRIDE + 4 * wave);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int co = co0 + h * 32 + 4 * wave + c;
            float R[4][2];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                R[x][0] = m[x * 4 + 0][c] + m[x * 4 + 1][c] + m[x * 4 + 2][c];
                R[x][1] = m[x * 4 + 1][c] - m[x * 4 + 2][c] - m[x * 4 + 3][c];
            }
            float Y[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                Y[0][jj] = R[0][jj] + R[1][jj] + R[2][jj];
                Y[1][jj] = R[1][jj] - R[2][jj] - R[3][jj];
            }
            if (co >= p.Cout) continue;
            const float b0 = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (oy + i >= p.OH || ox >= p.OW) continue;
                const int64_t o = (int64_t)co * out_plane + (int64_t)(oy + i) * p.OW + ox;
                const bool two = ox + 1 < p.OW;
                float v0 = Y[i][0] * inv + b0, v1 = Y[i][1] * inv + b0;
                if (pair_ok) {  // (even width: both pixels of the tile row exist)
                    if constexpr (ACC) {
                        const f32x2 prev = *reinterpret_cast<const f32x2*>(yout + o);
                        v0 += prev[0];
                        v1 += prev[1];
                    }
                    if (p.relu) {
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                    }
                    if constexpr (OM) {
                        const f32x2 mk = *reinterpret_cast<const f32x2*>(om + o);
                        v0 = mk[0] > 0.f ? v0 : 0.f;
                        v1 = mk[1] > 0.f ? v1 : 0.f;
                    }
                    const f32x2 out = {v0, v1};
                    *reinterpret_cast<f32x2*>(yout + o) = out;
                } else {
                    if constexpr (ACC) {
                        v0 += yout[o];
                        if (two) v1 += yout[o + 1];
                    }
                    if (p.relu) {
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                    }
                    if constexpr (OM) {
                        v0 = om[o] > 0.f ? v0 : 0.f;
                        if (two) v1 = om[o + 1] > 0.f ? v1 : 0.f;
                    }
                    yout[o] = v0;
                    if (two) yout[o + 1] = v1;
                }
            }
        }
    }
}

bool conv_wino_supports(int cin, int h, int w, int pad) {
    return cin % 16 == 0 && cin >= 16 && (int64_t)h * w <= (1ll << 25) && pad >= 0 && pad <= 2 && h + 2 * pad >= 3 && w + 2 * pad >= 3;
}

#ifdef WN_STAMP
static float* g_wn_stamp = nullptr;
extern "C" void maua_wn_set_stamp_buffer(float* buf) { g_wn_stamp = buf; }
#endif

extern "C" {

size_t maua_conv_wino_bank_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout > (1 << 20) || cin > (1 << 20)) return 0;  // (checked before any arithmetic on them)
    return (size_t)WN_BANK_HDR + (size_t)((cin + 15) / 16) * ((cout + WN_COT - 1) / WN_COT) * WN_U_BYTES;
}

int maua_conv_pack_filters_wino(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20), MAUA_E_INVAL,
                 "conv_pack_filters_wino: bad args");
    hipStream_t s = (hipStream_t)stream;
    for (int backward = 0; backward < 2; ++backward) {
        void* bank = backward ? bank_bwd : bank_fwd;
        if (!bank) continue;
        hipError_t e = hipMemsetAsync(bank, 0, WN_BANK_HDR, s);
        MAUA_REQUIRE(e == hipSuccess, (int)e, "conv_pack_filters_wino: %s", hipGetErrorString(e));
        const int64_t pairs = (int64_t)cout * cin;
        hipLaunchKernelGGL(wino_umax_kernel, dim3((unsigned)std::min<int64_t>((pairs + 255) / 256, 2048)), dim3(256), 0, s, w_oihw, cout, cin,
                           backward, (unsigned*)bank);
        const int CO = backward ? cin : cout, CI = backward ? cout : cin;
        const int64_t items = (int64_t)((CI + 15) / 16) * ((CO + WN_COT - 1) / WN_COT) * 128;
        hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)std::min<int64_t>((items + 127) / 128, 4096)), dim3(128), 0, s, w_oihw,
                           (unsigned char*)bank, cout, cin, backward);
    }
    return check_launch("wino_pack_kernel");
}

int maua_conv_wino_supported(int cin, int h, int w, int pad) {
    return conv_dims_ok(1, cin, h, w, 1, pad) && conv_wino_supports(cin, h, w, pad) ? 1 : 0;
}

int maua_conv3x3_wino(const float* x, const void* bank, const float* bias, const float* out_relu_mask, float* y, int n, int cin, int h,
                      int w, int cout, int pad, int relu, int accumulate, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y, MAUA_E_INVAL, "conv3x3_wino: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad), MAUA_E_INVAL, "conv3x3_wino: bad dims");
    MAUA_REQUIRE(conv_wino_supports(cin, h, w, pad), MAUA_E_UNSUPPORTED,
                 "conv3x3_wino: needs cin %% 16 == 0, pad <= 2 and a plane of at most 2^25 pixels");
    ConvArgs a{};
    a.x = x;
    a.w6 = bank;
    a.bias = bias;
    a.omask = out_relu_mask;
    a.y = y;
    a.Cin = cin;
    a.H = h;
    a.W = w;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    a.pad = pad;
    a.relu = relu;
    a.accumulate = accumulate;
    a.tiles_x = (a.OW + WN_COLS - 1) / WN_COLS;
#ifdef WN_STAMP
    a.mask = g_wn_stamp;
#endif
    const int ntile = (cout + WN_COT - 1) / WN_COT;
    const int tiles_sp = a.tiles_x * ((a.OH + WN_ROWS - 1) / WN_ROWS);
    int64_t blocks;
    if ((ntile & 7) == 0) blocks = (int64_t)ntile * tiles_sp;
    else if (ntile == 1 || ntile == 2 || ntile == 4) blocks = 8ll * ((tiles_sp + 8 / ntile - 1) / (8 / ntile));
    else blocks = (int64_t)ntile * tiles_sp;
    MAUA_REQUIRE(blocks < (1ll << 31) && n <= 65535, MAUA_E_UNSUPPORTED, "conv3x3_wino: grid too large");
    const dim3 grid((unsigned)blocks, 1, (unsigned)n), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool om = out_relu_mask != nullptr;
    if (accumulate) {
        if (om) hipLaunchKernelGGL((conv_wino_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_wino_kernel<true, false>), grid, block, 0, s, a);
    } else {
        if (om) hipLaunchKernelGGL((conv_wino_kernel<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_wino_kernel<false, false>), grid, block, 0, s, a);
    }
    return check_launch("conv_wino_kernel");
}

}  // extern "C"

}  // namespace maua
