// Direct (gather-form) convolution for geometry the MFMA path does not cover: strided filters such as NIN's
// 11x11 stride-4 stem (reference models.py:83).  One thread per output element; a wave covers 64 consecutive
// pixels of one channel, so the filter taps it needs are wave-uniform.  The forward reads the [tap][ci][co]
// bank, the backward-data reads the OIHW weights directly.
//
// Built WITHOUT packed fp32 instructions since round 6 (they are two plain instructions each: same bits).  No kernel here issues MFMAs, but
// the split-K finishing kernels run on the main stream right behind their convolution while a frame batch's per-frame Gram kernels (MFMA)
// may still be running on the side streams, and tools/soak_streams.py showed what round 5 only suspected: a packed-fp32 kernel beside an
// MFMA kernel of ANOTHER stream can lose results in the upper lanes (profiles/probes_r06.md section 2: conv3x3_few_out beside conv1x1_x3
// under the MFMA-padding build, 16951 of 17008 runs wrong, lanes 32-63).  conv3x3_few_out's packed FMAs (round 3: 108 -> 54 instructions
// per channel) go with it: it is the fallback of conv_few_mfma.hip now.
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include "common.hpp"

namespace maua {

__global__ void __launch_bounds__(256)
conv_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask, const float* __restrict__ wf,
                       const float* __restrict__ bias, float* __restrict__ y, int Cin, int H, int W, int Cout, int OH,
                       int OW, int KH, int KW, int stride, int pad, int relu, int accumulate) {
    const int64_t opix = (int64_t)OH * OW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y, n = blockIdx.z;
    if (idx >= opix) return;
    const int oy = (int)(idx / OW), ox = (int)(idx - (int64_t)oy * OW);
    const float* xin = x + (int64_t)n * Cin * H * W;
    const float* min_ = mask ? mask + (int64_t)n * Cin * H * W : nullptr;
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        for (int ky = 0; ky < KH; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < KW; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const int64_t a = ((int64_t)ci * H + iy) * W + ix;
                float v = xin[a];
                if (min_) v = min_[a] > 0.f ? v : 0.f;
                acc = fmaf(v, wf[((int64_t)(ky * KW + kx) * Cin + ci) * Cout + co], acc);
            }
        }
    }
    const int64_t o = ((int64_t)n * Cout + co) * opix + idx;
    if (accumulate) acc += y[o];
    if (relu) acc = acc > 0.f ? acc : 0.f;
    y[o] = acc;
}

__global__ void __launch_bounds__(256)
conv_direct_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ mask, const float* __restrict__ w_oihw,
                       const float* __restrict__ omask, float* __restrict__ gx, int Cin, int H, int W, int Cout, int OH,
                       int OW, int KH, int KW, int stride, int pad, int accumulate) {
    const int64_t ipix = (int64_t)H * W;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ci = blockIdx.y, n = blockIdx.z;
    if (idx >= ipix) return;
    const int iy = (int)(idx / W), ix = (int)(idx - (int64_t)iy * W);
    const float* g = gy + (int64_t)n * Cout * OH * OW;
    const float* m = mask ? mask + (int64_t)n * Cout * OH * OW : nullptr;
    float acc = 0.f;
    for (int ky = 0; ky < KH; ++ky) {
        const int ty = iy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= OH) continue;
        for (int kx = 0; kx < KW; ++kx) {
            const int tx = ix + pad - kx;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= OW) continue;
            for (int co = 0; co < Cout; ++co) {
                const int64_t a = ((int64_t)co * OH + oy) * OW + ox;
                float v = g[a];
                if (m) v = m[a] > 0.f ? v : 0.f;
                acc = fmaf(v, w_oihw[(((int64_t)co * Cin + ci) * KH + ky) * KW + kx], acc);
            }
        }
    }
    const int64_t o = ((int64_t)n * Cin + ci) * ipix + idx;
    if (accumulate) acc += gx[o];
    if (omask) acc = omask[o] > 0.f ? acc : 0.f;
    gx[o] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution that PRODUCES only a few channels (CO <= 4): the backward-data pass of conv1_1
// (64 gradient channels -> 3 pixel channels, reference models.py:129 through autograd).  A 64-row MFMA tile would be
// 95 % padding; the work is 27 FMAs per input value, so this is a vector-ALU kernel bound by reading the input once:
// lane = column, four output rows per thread, the left/right taps come from the neighbouring lanes through DPP
// wave shifts (lanes 0 and 63 are halo), the 27 filter values of a channel are wave-uniform (scalar registers).
// ---------------------------------------------------------------------------------------------------------
constexpr int FO_TW = 62;
typedef float fo_f2 __attribute__((ext_vector_type(2)));

__device__ inline float lane_from_left(float v) {  // lane i <- lane i-1  (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ inline float lane_from_right(float v) {  // lane i <- lane i+1  (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// KS4 (small images: a 256 x 256 plane is 80 workgroups of the plain form on 256 CUs, each walking all 64 channels): the four waves of
// a workgroup share ONE strip of FO_R rows and take a quarter of the input channels each; waves 1-3 leave their sums in LDS and wave 0
// adds them in wave order (fixed order).  Four times the workgroups, a quarter of the dependent channel steps per wave.
template <int CO, bool MASK, bool KS4 = false, int FO_R = KS4 ? 8 : 4>  // rows per thread: 4 (plain: 83.7 vs 89.6 us at 1024^2 with 8), 8 (KS4: 31.5 vs 34.9 us at 512^2)
__global__ void __launch_bounds__(256)
conv3x3_few_out_kernel(ConvArgs p, const float* __restrict__ wbank) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.z;
    const int ox = blockIdx.x * FO_TW + lane - 1;          // output column (lanes 1..62 are stored)
    const int oy0 = (KS4 ? blockIdx.y : blockIdx.y * 4 + wave) * FO_R;  // first of the FO_R output rows of this wave
    const int ci_begin = KS4 ? wave * (p.Cin >> 2) : 0, ci_end = KS4 ? ci_begin + (p.Cin >> 2) : p.Cin;
    const int cx = ox - p.pad + 1;                         // input column of the centre tap
    const bool col_ok = cx >= 0 && cx < p.W;
    const int64_t plane = (int64_t)p.H * p.W;
    int off[FO_R + 2];
    bool ok[FO_R + 2];
#pragma unroll
    for (int r = 0; r < FO_R + 2; ++r) {
        const int iy = oy0 - p.pad + r;
        ok[r] = col_ok && iy >= 0 && iy < p.H;
        off[r] = min(max(iy, 0), p.H - 1) * p.W + min(max(cx, 0), p.W - 1);
        asm volatile("" : "+v"(off[r]));  // keep the loads unconditional: the select below zeroes the padding
    }
    const float* xin = p.x + (int64_t)n * p.Cin * plane;
    const float* min_ = MASK ? p.mask + (int64_t)n * p.Cin * plane : nullptr;
    auto load = [&](int ci, float (&v)[FO_R + 2]) {
        const float* xc = xin + (int64_t)ci * plane;
#pragma unroll
        for (int r = 0; r < FO_R + 2; ++r) {
            float t = xc[off[r]];
            if (MASK) t = min_[(int64_t)ci * plane + off[r]] > 0.f ? t : 0.f;
            v[r] = ok[r] ? t : 0.f;
        }
    };
    float master[FO_R][CO], acc[FO_R][CO];
#pragma unroll
    for (int r = 0; r < FO_R; ++r)
#pragma unroll
        for (int c = 0; c < CO; ++c) master[r][c] = acc[r][c] = 0.f;
    float cur[FO_R + 2], nxt[FO_R + 2];
    load(ci_begin, cur);
    for (int ci = ci_begin; ci < ci_end; ++ci) {
        load(min(ci + 1, ci_end - 1), nxt);  // unconditional (the last one re-reads its own channel)
        float wv[9][CO];  // wave-uniform: bank layout [tap][ci][co]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < CO; ++c) wv[t][c] = wbank[((int64_t)t * p.Cin + ci) * CO + c];
        // packed fp32 FMAs (v_pk_fma_f32: two rows per instruction, the filter value broadcast from a scalar register): tap row ky takes
        // input rows (2 p + ky, 2 p + ky + 1) to output rows (2 p, 2 p + 1) - 54 packed instructions per channel instead of 108 FMAs
        float lf[FO_R + 2], rg[FO_R + 2];
#pragma unroll
        for (int r = 0; r < FO_R + 2; ++r) {
            lf[r] = lane_from_left(cur[r]);
            rg[r] = lane_from_right(cur[r]);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int pr = 0; pr < FO_R / 2; ++pr) {
                const fo_f2 l2 = {lf[2 * pr + ky], lf[2 * pr + ky + 1]}, m2 = {cur[2 * pr + ky], cur[2 * pr + ky + 1]},
                            r2 = {rg[2 * pr + ky], rg[2 * pr + ky + 1]};
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    fo_f2 a2 = {acc[2 * pr][c], acc[2 * pr + 1][c]};
                    a2 = __builtin_elementwise_fma(l2, (fo_f2){wv[ky * 3 + 0][c], wv[ky * 3 + 0][c]}, a2);
                    a2 = __builtin_elementwise_fma(m2, (fo_f2){wv[ky * 3 + 1][c], wv[ky * 3 + 1][c]}, a2);
                    a2 = __builtin_elementwise_fma(r2, (fo_f2){wv[ky * 3 + 2][c], wv[ky * 3 + 2][c]}, a2);
                    acc[2 * pr][c] = a2.x;
                    acc[2 * pr + 1][c] = a2.y;
                }
            }
        if ((ci & 7) == 7 || ci + 1 == ci_end) {  // two-level accumulation like the MFMA kernels
#pragma unroll
            for (int r = 0; r < FO_R; ++r)
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    master[r][c] += acc[r][c];
                    acc[r][c] = 0.f;
                }
        }
#pragma unroll
        for (int r = 0; r < FO_R + 2; ++r) cur[r] = nxt[r];
    }
    if constexpr (KS4) {
        __shared__ float red[3][FO_R * CO][64];
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < FO_R; ++r)
#pragma unroll
                for (int c = 0; c < CO; ++c) red[wave - 1][r * CO + c][lane] = master[r][c];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int r = 0; r < FO_R; ++r)
#pragma unroll
                for (int c = 0; c < CO; ++c) master[r][c] += red[w][r * CO + c][lane];
    }
    if (lane < 1 || lane > FO_TW || ox >= p.OW) return;
    const int64_t oplane = (int64_t)p.OH * p.OW;
#pragma unroll
    for (int r = 0; r < FO_R; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.OH) break;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const int64_t o = ((int64_t)n * CO + c) * oplane + (int64_t)oy * p.OW + ox;
            float v = master[r][c];
            if (p.bias) v += p.bias[c];
            if (p.accumulate) v += p.y[o];
            if (p.relu) v = v > 0.f ? v : 0.f;
            if (p.omask) v = p.omask[o] > 0.f ? v : 0.f;
            p.y[o] = v;
        }
    }
}

// out = act(bias + sum_split ws[split]) (+ out) masked: the fixed-order second stage of the split-K convolution
__global__ void __launch_bounds__(256)
conv_splitk_finish_kernel(const float* __restrict__ ws, const float* __restrict__ bias, const float* __restrict__ omask,
                          float* __restrict__ y, int ksplit, int Cout, int64_t out_plane, int relu, int accumulate) {
    // grid = (pixel quads of a plane / 256, Cout, n): no index divisions, the channel's bias is wave-uniform, 16-byte
    // accesses (dword alignment is enough in global memory, so odd-sized planes take them too; a plane's last, partial
    // quad goes element by element); the ksplit partials are added in index order (deterministic)
    typedef float v4 __attribute__((ext_vector_type(4), aligned(4)));
    const int co = blockIdx.y, n = blockIdx.z;
    const int64_t per_n = (int64_t)Cout * out_plane;
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= out_plane) return;
    const int64_t r = (int64_t)co * out_plane + p, e = (int64_t)n * per_n + r;
    const float* src = ws + (int64_t)n * ksplit * per_n + r;
    const float b = bias ? bias[co] : 0.f;
    if (p + 4 <= out_plane) {
        v4 v = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ksplit; ++k) v += *reinterpret_cast<const v4*>(src + (int64_t)k * per_n);
        v += b;
        if (accumulate) v += *reinterpret_cast<const v4*>(y + e);
        v4 m = {1.f, 1.f, 1.f, 1.f};
        if (omask) m = *reinterpret_cast<const v4*>(omask + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = v[i];
            if (relu) t = t > 0.f ? t : 0.f;
            v[i] = m[i] > 0.f ? t : 0.f;
        }
        *reinterpret_cast<v4*>(y + e) = v;
    } else {
        for (int i = 0; p + i < out_plane; ++i) {
            float v = 0.f;
            for (int k = 0; k < ksplit; ++k) v += src[(int64_t)k * per_n + i];
            v += b;
            if (accumulate) v += y[e + i];
            if (relu) v = v > 0.f ? v : 0.f;
            if (omask) v = omask[e + i] > 0.f ? v : 0.f;
            y[e + i] = v;
        }
    }
}

int conv_splitk_finish(const ConvArgs& a, int n, int ksplit, hipStream_t stream) {
    const int64_t out_plane = (int64_t)a.OH * a.OW;
    dim3 grid((unsigned)(((out_plane + 3) / 4 + 255) / 256), (unsigned)a.Cout, (unsigned)n);
    hipLaunchKernelGGL(conv_splitk_finish_kernel, grid, dim3(256), 0, stream, a.ws, a.bias, a.omask, a.y, ksplit, a.Cout, out_plane,
                       a.relu, a.accumulate);
    return check_launch("conv_splitk_finish_kernel");
}

// The same second stage for a convolution whose ReLU output feeds nothing but a 2x2 / 2 max pool (conv_x3w's pooling epilogue needs
// complete sums, which a split channel loop does not have): relu(bias + sum of the slabs) of a window's four elements - the additions in
// conv_splitk_finish_kernel's order - then pool2x2_fwd_codes_kernel's decision (first maximum in scan order, NaN wins; bit 2 = not
// positive) and its byte layout.  The full-size activation is not written.
template <bool ALIGNED>  // (even output width: a window row is one 8-byte load; odd planes - floor-mode pooling - drop their last row / column)
__global__ void __launch_bounds__(256)
conv_splitk_finish_pool_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ pooled,
                               unsigned char* __restrict__ codes, int ksplit, int Cout, int OW, int PW, int64_t pplane, int64_t out_plane) {
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y;
    if (px >= PW) return;
    const int nc = blockIdx.z, n = nc / Cout, co = nc - n * Cout;  // (image, channel): one division per workgroup
    const int64_t per_n = (int64_t)Cout * out_plane;
    const float* src = ws + (int64_t)n * ksplit * per_n + (int64_t)co * out_plane + (int64_t)(2 * py) * OW + 2 * px;
    float2 r0 = make_float2(0.f, 0.f), r1 = make_float2(0.f, 0.f);
    for (int k = 0; k < ksplit; ++k) {
        float2 a, c;
        if constexpr (ALIGNED) {
            a = *reinterpret_cast<const float2*>(src + (int64_t)k * per_n);
            c = *reinterpret_cast<const float2*>(src + (int64_t)k * per_n + OW);
        } else {
            a = make_float2(src[(int64_t)k * per_n], src[(int64_t)k * per_n + 1]);
            c = make_float2(src[(int64_t)k * per_n + OW], src[(int64_t)k * per_n + OW + 1]);
        }
        r0.x += a.x; r0.y += a.y; r1.x += c.x; r1.y += c.y;
    }
    const float b = bias ? bias[co] : 0.f;
    r0.x += b; r0.y += b; r1.x += b; r1.y += b;
    r0.x = r0.x > 0.f ? r0.x : 0.f; r0.y = r0.y > 0.f ? r0.y : 0.f; r1.x = r1.x > 0.f ? r1.x : 0.f; r1.y = r1.y > 0.f ? r1.y : 0.f;
    float m = r0.x;
    int arg = 0;
    if (r0.y > m || r0.y != r0.y) { m = r0.y; arg = 1; }
    if (r1.x > m || r1.x != r1.x) { m = r1.x; arg = 2; }
    if (r1.y > m || r1.y != r1.y) { m = r1.y; arg = 3; }
    const int64_t ppix = (int64_t)py * PW + px;
    pooled[(int64_t)nc * pplane + ppix] = m;
    codes[(((int64_t)nc >> 3) * pplane + ppix) * 8 + (nc & 7)] = (unsigned char)(arg | (m > 0.f ? 0 : 4));  // (Cout % 8 == 0)
}

int conv_splitk_finish_pool(const ConvArgs& a, int n, int ksplit, hipStream_t stream) {
    const int PW = a.OW / 2, PH = a.OH / 2;
    dim3 grid((unsigned)((PW + 255) / 256), (unsigned)PH, (unsigned)(n * a.Cout));
    if (a.OW % 2 == 0)
        hipLaunchKernelGGL(conv_splitk_finish_pool_kernel<true>, grid, dim3(256), 0, stream, a.ws, a.bias, a.y, a.pool_codes, ksplit, a.Cout, a.OW,
                           PW, (int64_t)PW * PH, (int64_t)a.OH * a.OW);
    else
        hipLaunchKernelGGL(conv_splitk_finish_pool_kernel<false>, grid, dim3(256), 0, stream, a.ws, a.bias, a.y, a.pool_codes, ksplit, a.Cout, a.OW,
                           PW, (int64_t)PW * PH, (int64_t)a.OH * a.OW);
    return check_launch("conv_splitk_finish_pool_kernel");
}

int conv3x3_few_out(const ConvArgs& a, int n, hipStream_t stream) {
    dim3 grid((unsigned)((a.OW + FO_TW - 1) / FO_TW), (unsigned)((a.OH + 4 * 4 - 1) / (4 * 4)), (unsigned)n);  // (4 waves x 4 rows)
    // channel quarters per wave where the plain grid leaves most of the chip idle (the planned frames per launch count, not this
    // launch's: a frame's bits do not depend on how many others share its launch)
    const int ks4_below = (int)tuning("few_out_ks4_below", 1024);
    const bool ks4 = (int64_t)grid.x * grid.y * split_batch_hint() < ks4_below && a.Cin % 32 == 0;
    if (ks4) grid.y = (unsigned)((a.OH + 8 - 1) / 8);  // (one strip of 8 rows per workgroup)
#define MAUA_FO(CO_)                                                                                            \
    case CO_:                                                                                                   \
        if (ks4 && a.mask) hipLaunchKernelGGL((conv3x3_few_out_kernel<CO_, true, true>), grid, dim3(256), 0, stream, a, a.w);  \
        else if (ks4) hipLaunchKernelGGL((conv3x3_few_out_kernel<CO_, false, true>), grid, dim3(256), 0, stream, a, a.w);      \
        else if (a.mask) hipLaunchKernelGGL((conv3x3_few_out_kernel<CO_, true>), grid, dim3(256), 0, stream, a, a.w); \
        else hipLaunchKernelGGL((conv3x3_few_out_kernel<CO_, false>), grid, dim3(256), 0, stream, a, a.w);             \
        break;
    switch (a.Cout) {
        MAUA_FO(1) MAUA_FO(2) MAUA_FO(3) MAUA_FO(4)
        default: MAUA_REQUIRE(false, MAUA_E_UNSUPPORTED, "conv3x3_few_out: %d output channels", a.Cout);
    }
#undef MAUA_FO
    return check_launch("conv3x3_few_out_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// Strided stem (NIN conv1: 11x11 stride 4, 3 -> 96 channels, no padding; reference models.py:83).  Both passes are
// vector-ALU kernels with wave-uniform filter values (scalar loads, no LDS):
//   forward:  thread = one output pixel x COG output channels; a filter row's KS input values are contiguous
//             (unaligned vector loads straight from global memory, L1 absorbs the overlap between neighbours).
//   backward: thread = S consecutive input pixels of one row (all S column phases) x all CI input channels.  With
//             ky = (iy mod S) + S a and kx = px + S b every filter value is used exactly once per thread and the
//             gradient values it multiplies are the 3x3 neighbourhood gy[co][iy/S - a][ix/S - b].
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int KS, int S, int COG>
__global__ void __launch_bounds__(256)
conv_strided_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bias,
                        float* __restrict__ y, int Cin, int H, int W, int Cout, int OH, int OW, int relu, int accumulate) {
    const int64_t opix = (int64_t)OH * OW;
    const int64_t idx_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t idx = idx_raw < opix ? idx_raw : opix - 1;  // out-of-range threads recompute the last pixel, store nothing
    const int co0 = blockIdx.y * COG, n = blockIdx.z;
    const int oy = (int)(idx / OW), ox = (int)(idx - (int64_t)oy * OW);
    float acc[COG];
#pragma unroll
    for (int c = 0; c < COG; ++c) acc[c] = 0.f;
    const float* xin = x + (int64_t)n * Cin * H * W + (int64_t)(oy * S) * W + ox * S;
    for (int ci = 0; ci < Cin; ++ci) {
        for (int ky = 0; ky < KS; ++ky) {
            const float* row = xin + ((int64_t)ci * H + ky) * W;
            float v[KS];
#pragma unroll
            for (int q = 0; q + 4 <= KS; q += 4) {
                const f32x4u t = *reinterpret_cast<const f32x4u*>(row + q);
                v[q] = t[0], v[q + 1] = t[1], v[q + 2] = t[2], v[q + 3] = t[3];
            }
#pragma unroll
            for (int q = KS & ~3; q < KS; ++q) v[q] = row[q];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const float* wr = wf + ((int64_t)(ky * KS + kx) * Cin + ci) * Cout + co0;  // wave-uniform
#pragma unroll
                for (int c = 0; c < COG; ++c) acc[c] = fmaf(v[kx], wr[c], acc[c]);
            }
        }
    }
    if (idx_raw >= opix) return;
#pragma unroll
    for (int c = 0; c < COG; ++c) {
        const int64_t o = ((int64_t)n * Cout + co0 + c) * opix + idx;
        float r = acc[c] + (bias ? bias[co0 + c] : 0.f);
        if (accumulate) r += y[o];
        if (relu) r = r > 0.f ? r : 0.f;
        y[o] = r;
    }
}

template <int KS, int S, int CI>
__global__ void __launch_bounds__(256)
conv_strided_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ w_oihw, const float* __restrict__ omask,
                        float* __restrict__ gx, int H, int W, int Cout, int OH, int OW, int accumulate) {
    constexpr int NA = (KS + S - 1) / S;  // filter rows / columns per phase
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;  // block of S input columns
    const int iy = blockIdx.y, n = blockIdx.z;
    const int py = iy % S, by = iy / S;                   // wave-uniform
    float acc[CI][S];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int px = 0; px < S; ++px) acc[ci][px] = 0.f;
    const float* g = gy + (int64_t)n * Cout * OH * OW;
    // clamped gather offsets + validity of the NA x NA neighbourhood (same for every co)
    int goff[NA][NA];
    bool gok[NA][NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NA; ++b) {
            const int oy = by - a, ox = bx - b;
            gok[a][b] = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
            goff[a][b] = min(max(oy, 0), OH - 1) * OW + min(max(ox, 0), OW - 1);
        }
    for (int co = 0; co < Cout; ++co) {
        const float* gc = g + (int64_t)co * OH * OW;
        float gv[NA][NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) {
                const float t = gc[goff[a][b]];
                gv[a][b] = gok[a][b] ? t : 0.f;
            }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int ky = py + S * a;
            if (ky >= KS) continue;  // wave-uniform
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float* wr = w_oihw + (((int64_t)co * CI + ci) * KS + ky) * KS;  // wave-uniform row of KS values
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) acc[ci][kx % S] = fmaf(gv[a][kx / S], wr[kx], acc[ci][kx % S]);
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int px = 0; px < S; ++px) {
            const int ix = bx * S + px;
            if (ix >= W) continue;
            const int64_t o = (((int64_t)n * CI + ci) * H + iy) * W + ix;
            float r = acc[ci][px];
            if (accumulate) r += gx[o];
            if (omask) r = omask[o] > 0.f ? r : 0.f;
            gx[o] = r;
        }
}

int conv_direct_fwd(const float* x, const float* mask, const float* wf, const float* bias, float* y, int n, int cin, int h,
                    int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int relu, int accumulate,
                    hipStream_t stream) {
    if (kh == 11 && kw == 11 && stride == 4 && pad == 0 && !mask && cout % 16 == 0) {
        dim3 sgrid((unsigned)(((int64_t)oh * ow + 255) / 256), (unsigned)(cout / 16), (unsigned)n);
        hipLaunchKernelGGL((conv_strided_fwd_kernel<11, 4, 16>), sgrid, dim3(256), 0, stream, x, wf, bias, y, cin, h, w, cout,
                           oh, ow, relu, accumulate);
        return check_launch("conv_strided_fwd_kernel");
    }
    dim3 grid((unsigned)(((int64_t)oh * ow + 255) / 256), (unsigned)cout, (unsigned)n);
    hipLaunchKernelGGL(conv_direct_fwd_kernel, grid, dim3(256), 0, stream, x, mask, wf, bias, y, cin, h, w, cout, oh, ow,
                       kh, kw, stride, pad, relu, accumulate);
    return check_launch("conv_direct_fwd_kernel");
}

int conv_direct_bwd(const float* gy, const float* mask, const float* w_oihw, const float* omask, float* gx, int n, int cin,
                    int h, int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int accumulate,
                    hipStream_t stream) {
    if (kh == 11 && kw == 11 && stride == 4 && pad == 0 && !mask && cin == 3) {
        const int blocks_x = (w + 3) / 4;
        dim3 sgrid((unsigned)((blocks_x + 255) / 256), (unsigned)h, (unsigned)n);
        hipLaunchKernelGGL((conv_strided_bwd_kernel<11, 4, 3>), sgrid, dim3(256), 0, stream, gy, w_oihw, omask, gx, h, w, cout,
                           oh, ow, accumulate);
        return check_launch("conv_strided_bwd_kernel");
    }
    dim3 grid((unsigned)(((int64_t)h * w + 255) / 256), (unsigned)cin, (unsigned)n);
    hipLaunchKernelGGL(conv_direct_bwd_kernel, grid, dim3(256), 0, stream, gy, mask, w_oihw, omask, gx, cin, h, w, cout,
                       oh, ow, kh, kw, stride, pad, accumulate);
    return check_launch("conv_direct_bwd_kernel");
}

}  // namespace maua
