// 1x1 convolution / channel-mixing GEMM  y[co][p] = sum_ci w[co][ci] x[ci][p]  at fp32-level accuracy on the fp16 matrix
// cores (the "fp16x3" arithmetic of conv_x3.hip: two fp16 parts per operand, products hh + hl + lh, fp32 accumulate).
// Users: NIN's 1x1 layers (reference models.py:84-110, forward and backward-data) and the Gram backward
// gf (+)= D (F - mean) of loss.py:91's autograd, whose weight matrix D changes every iteration.
//
// Both operands are plain fp32 tensors in HBM: the 32-channel chunk of activations (128 pixels x 32 channels) AND the
// matching 64 x 32 block of weights are scaled (power of two, from the workgroup's maxima of exactly these values) and
// split while they are staged into LDS, so there is no pre-packed bank and nothing to re-pack when the weights change.
// Workgroup = 4 waves = 64 output channels x 128 pixels, wave = 64 x 32 (two accumulators + two fp32 master accumulators
// that collect every chunk's sums un-scaled); chunk = 32 input channels = 4 groups of 8 = two K=16 MFMA steps (lane half
// h takes group 2s+h).  LDS: activations [part][group][pixel][8ch] 16 KB + weights [group][part][co][8ch] 8 KB, both
// conflict-free for b128.  Without the tap reuse of a 3x3 filter this is a bandwidth / staging-bound kernel (2 MFMA
// steps per 16 staged values per thread); what it buys is the fp32 matrix cores' 16x lower rate out of the way.

// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int P1_COT = 64, P1_PX = 128, P1_KC = 32;
constexpr int P1_X_BYTES = 2 * 4 * P1_PX * 16;   // 16384
constexpr int P1_W_BYTES = 4 * 2 * P1_COT * 16;  // 8192

__device__ __forceinline__ unsigned p1_cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float p1_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float p1_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
// scale that brings `m` into [2^target, 2^(target+1)); returns the scale, *inv = its inverse (both powers of two)
__device__ __forceinline__ float p1_scale(float m, int target, float* inv) {
    int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
    e = m > 0.f ? max(e, -100) : target;
    *inv = __builtin_bit_cast(float, (unsigned)(127 + e - target) << 23);
    return __builtin_bit_cast(float, (unsigned)(127 + target - e) << 23);
}

// a.x = activations [n][Cin][HW], a.w = weights [Cout][Cin] row-major, a.H * a.W = HW, a.OH * a.OW = HW
// xshift (SH): per-input-channel value subtracted from x while staging (the Gram backward's centring, loss.py:80)
template <bool ACC, bool OM, bool SH>
__global__ void __launch_bounds__(256, 4) conv1x1_x3_kernel(ConvArgs p, const float* __restrict__ xshift) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[P1_X_BYTES + P1_W_BYTES + 32];
    unsigned char* Xl = smem;                // [part][group][pixel][16 B]
    unsigned char* Wl = smem + P1_X_BYTES;   // [group][part][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + P1_X_BYTES + P1_W_BYTES);  // maxima: [0..3] activations, [4..7] weights (per wave)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    const int64_t HW = (int64_t)p.H * p.W;
    // Tile order (round 5): with p.tiles_x = items per XCD the (pixel tile, channel tile) items - channel tiles of one pixel tile next to
    // each other - are dealt so that XCD k (workgroup ids = k mod 8) walks the k-th contiguous band of the list, its CUs side by side: the
    // channel tiles of a pixel tile share its activations in that XCD's L2, neighbouring CUs read neighbouring row segments of the same
    // planes at the same time (dispatch order - p.tiles_x = 0: grid (pixel tiles, channel tiles) - reads the same bytes at a lower rate:
    // tools/mfma_probe/stage_bw.hip, conv_few_mfma.hip).
    int ptile = blockIdx.x, cot = blockIdx.y;
    if (p.tiles_x > 0) {
        const int ncot = (p.Cout + P1_COT - 1) / P1_COT;
        const int item = (int)(blockIdx.x & 7) * p.tiles_x + (int)(blockIdx.x >> 3);
        ptile = item / ncot;
        cot = item - ptile * ncot;
        if ((int64_t)ptile * P1_PX >= HW) return;  // (whole workgroup)
    }
    const int co0 = cot * P1_COT;
    const int64_t pix0 = (int64_t)ptile * P1_PX;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * HW;
    const float* __restrict__ wgt = p.w;

    // staging descriptors: activations - pixel tid & 127, groups sg and sg + 2 (sg = tid >> 7 is wave-uniform); weights -
    // output channel tid & 63, group wg = wave.  Out-of-range pixels / output channels re-read valid ones (their results
    // are never stored), channels past Cin re-read channel Cin - 1 and meet zeroed weights: no per-value masks.
    const int spx = tid & 127;
    const int sg = __builtin_amdgcn_readfirstlane(tid >> 7), wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned x_byte = (unsigned)min(pix0 + spx, HW - 1) * 4u;  // HW < 2^30 (checked by the entry point)
    const int wco = tid & 63;
    const unsigned w_byte = (unsigned)min(co0 + wco, p.Cout - 1) * (unsigned)p.Cin * 4u;  // Cout * Cin < 2^30 (checked)

    f32x2 rx[2][4], rw[4];
    float sh[2][8];  // wave-uniform
    const int64_t plane_bytes = HW * 4;
    auto load_chunk = [&](int c0) {
        asm volatile("" : "+s"(c0));
        if (c0 + P1_KC <= p.Cin) {  // whole chunk in range: one scalar base, running plane pointer
            const char* plane = reinterpret_cast<const char*>(xin + (int64_t)(c0 + sg * 8) * HW);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    rx[s][c >> 1][c & 1] = *reinterpret_cast<const float*>(plane + x_byte);
                    plane += plane_bytes;
                    if constexpr (SH) sh[s][c] = xshift[c0 + (sg + 2 * s) * 8 + c];
                }
                plane += 8 * plane_bytes;
            }
            const char* col = reinterpret_cast<const char*>(wgt + c0 + wg * 8);
#pragma unroll
            for (int c = 0; c < 8; ++c) rw[c >> 1][c & 1] = *reinterpret_cast<const float*>(col + w_byte + 4 * c);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int chn = min(c0 + (sg + 2 * s) * 8 + c, p.Cin - 1);
                    const char* plane = reinterpret_cast<const char*>(xin + (int64_t)chn * HW);
                    rx[s][c >> 1][c & 1] = *reinterpret_cast<const float*>(plane + x_byte);
                    if constexpr (SH) sh[s][c] = xshift[chn];
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int chn = c0 + wg * 8 + c;
                const char* col = reinterpret_cast<const char*>(wgt + min(chn, p.Cin - 1));
                const float v = *reinterpret_cast<const float*>(col + w_byte);
                rw[c >> 1][c & 1] = chn < p.Cin ? v : 0.f;
            }
        }
    };
    auto publish_max = [&]() {
        float mx = 0.f, mw = 0.f;
        if constexpr (SH) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) rx[s][c >> 1][c & 1] -= sh[s][c];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) mx = fmaxf(fmaxf(mx, fabsf(rx[s][q][0])), fabsf(rx[s][q][1]));
#pragma unroll
        for (int q = 0; q < 4; ++q) mw = fmaxf(fmaxf(mw, fabsf(rw[q][0])), fabsf(rw[q][1]));
        mx = wave_max_nonneg(mx);
        mw = wave_max_nonneg(mw);
        if (lane == 0) {
            Ml[wave] = mx;
            Ml[4 + wave] = mw;
        }
    };
    auto split_pair = [&](f32x2 v, unsigned& hi, unsigned& lo) {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
        const f32x2 back = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, f16x2));
    };
    // returns the factor that un-scales the chunk's products; stages the chunk held in rx / rw
    auto stage_chunk = [&]() {
        float ix, iw;
        const float sx = p1_scale(fmaxf(fmaxf(Ml[0], Ml[1]), fmaxf(Ml[2], Ml[3])), 11, &ix);
        const float sw = p1_scale(fmaxf(fmaxf(Ml[4], Ml[5]), fmaxf(Ml[6], Ml[7])), 5, &iw);
        const f32x2 sx2 = {sx, sx}, sw2 = {sw, sw};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 vh, vl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned H, L;
                split_pair(rx[s][q] * sx2, H, L);
                vh[q] = H;
                vl[q] = L;
            }
            const int g = sg + 2 * s;
            *reinterpret_cast<u32x4*>(Xl + ((0 * 4 + g) * P1_PX + spx) * 16) = vh;
            *reinterpret_cast<u32x4*>(Xl + ((1 * 4 + g) * P1_PX + spx) * 16) = vl;
        }
        {
            u32x4 vh, vl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned H, L;
                split_pair(rw[q] * sw2, H, L);
                vh[q] = H;
                vl[q] = L;
            }
            *reinterpret_cast<u32x4*>(Wl + ((wg * 2 + 0) * P1_COT + wco) * 16) = vh;
            *reinterpret_cast<u32x4*>(Wl + ((wg * 2 + 1) * P1_COT + wco) * 16) = vl;
        }
        return ix * iw;
    };

    f32x16 acc[2], master[2];
    {
        const bool with_bias = p.bias != nullptr && p.ksplit <= 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float b0 = 0.f;
                if (with_bias) b0 = p.bias[min(co, p.Cout - 1)];
                master[t][r] = b0;
            }
    }
    auto kstep = [&](int s) {  // step 0 starts the chunk's sums from zero (constant C operand: no register clearing)
        const int g = 2 * s + half;
        f16x8 b[2], a[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) b[part] = *reinterpret_cast<const f16x8*>(Xl + ((part * 4 + g) * P1_PX + wave * 32 + j) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                a[t][part] = *reinterpret_cast<const f16x8*>(Wl + ((g * 2 + part) * P1_COT + t * 32 + j) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[0], s == 0 ? zero : acc[t], 0, 0, 0);  // smallest terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };

    const int nchunks_all = (p.Cin + P1_KC - 1) / P1_KC;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);
    load_chunk(ch_begin * P1_KC);
    publish_max();
    __syncthreads();
    float inv_cur = stage_chunk();
    __syncthreads();
    for (int ch = ch_begin; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * P1_KC);
        kstep(0);
        kstep(1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                master[t][r] = fmaf(acc[t][r], inv_cur, master[t][r]);  // fold + un-scale (power of two: exact)
            }
        if (more) publish_max();
        __syncthreads();  // every wave is done reading the chunk; the maxima of the next one are visible
        if (more) {
            inv_cur = stage_chunk();
            __syncthreads();
        }
    }

    // epilogue: lane holds pixel pix0 + wave*32 + j; register r is output channel (r&3)+8*(r>>2)+4*half of block t
    const int64_t opix = pix0 + wave * 32 + j;
    const bool pvalid = opix < HW;
    if (p.ksplit > 1) {
        float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * HW;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pvalid && co < p.Cout) wsp[(int64_t)co * HW + opix] = master[t][r];
            }
        return;
    }
    if (pvalid) {
        const bool full = co0 + P1_COT <= p.Cout;
        const int64_t lane_off = ((int64_t)n * p.Cout + co0 + 4 * half) * HW + opix;
        float* __restrict__ yl = p.y + lane_off;
        const float* __restrict__ oml = OM ? p.omask + lane_off : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float prev[16], msk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                const int64_t o = (full || co0 + cr + 4 * half < p.Cout) ? (int64_t)cr * HW : 0;
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = yl[o];
                if constexpr (OM) msk[r] = oml[o];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                if (full || co0 + cr + 4 * half < p.Cout) {
                    float v = master[t][r] + prev[r];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    yl[(int64_t)cr * HW] = msk[r] > 0.f ? v : 0.f;
                }
            }
        }
    }
}

static int p1_choose_split(const ConvArgs& a, int n) {
    const int64_t hw = (int64_t)a.H * a.W;
    const int64_t wgs = ((hw + P1_PX - 1) / P1_PX) * ((a.Cout + P1_COT - 1) / P1_COT) * split_batch_hint();  // planned frames, not this launch's (conv_x3w.hip)
    const int nchunks = (a.Cin + P1_KC - 1) / P1_KC;
    if (wgs >= 2048 || nchunks < 8) return 1;
    const double out_mb = (double)n * a.Cout * hw * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 4; ++ks) {
        const double rounds = (double)((wgs * ks + 1023) / 1024);
        double cost = rounds * ((double)((nchunks + ks - 1) / ks) + 2.0) * 1.5;
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

// a: x, w ([Cout][Cin] row-major), bias, omask, y, Cin, Cout, H*W = pixels per plane, relu, accumulate, ws;
// y (+)= w (x - xshift[channel]) + bias
int conv1x1_x3_launch(const ConvArgs& a, const float* xshift, int n, hipStream_t stream) {
    ConvArgs p = a;
    p.OH = a.H;
    p.OW = a.W;
    const int64_t hw = (int64_t)a.H * a.W;
    const int ks = a.ws ? p1_choose_split(a, n) : 1;
    p.ksplit = ks;
    dim3 grid((unsigned)((hw + P1_PX - 1) / P1_PX), (unsigned)((a.Cout + P1_COT - 1) / P1_COT), (unsigned)(n * ks));
    p.tiles_x = 0;
    if (tuning("p1_order", 1) != 0) {
        const int64_t items = (int64_t)grid.x * grid.y;
        p.tiles_x = (int)((items + 7) / 8);
        grid = dim3((unsigned)(8 * p.tiles_x), 1, (unsigned)(n * ks));
    }
    const bool acc = ks == 1 && a.accumulate != 0, om = ks == 1 && a.omask != nullptr;
#define MAUA_P1(ACC_, OM_)                                                                                                   \
    do {                                                                                                                     \
        if (xshift) hipLaunchKernelGGL((conv1x1_x3_kernel<ACC_, OM_, true>), grid, dim3(256), 0, stream, p, xshift);          \
        else hipLaunchKernelGGL((conv1x1_x3_kernel<ACC_, OM_, false>), grid, dim3(256), 0, stream, p, xshift);                \
    } while (0)
    if (acc && om) MAUA_P1(true, true);
    else if (acc) MAUA_P1(true, false);
    else if (om) MAUA_P1(false, true);
    else MAUA_P1(false, false);
#undef MAUA_P1
    int rc = check_launch("conv1x1_x3_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(p, n, ks, stream);
}

size_t conv1x1_x3_workspace(int n, int cin, int64_t hw, int cout) {
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.H = 1;
    a.W = (int)hw;
    const int ks = p1_choose_split(a, n);
    return ks > 1 ? (size_t)n * ks * cout * hw * sizeof(float) : 0;
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv1x1_x3_workspace_bytes(int n, int cin, int64_t hw, int cout) {
    if (!conv_dims_ok(n, cin, 1, hw < (1 << 24) ? hw : 1, cout, 0) || hw <= 0 || hw >= (1ll << 30)) return 0;
    return conv1x1_x3_workspace(n, cin, hw, cout);
}

int maua_conv1x1_x3(const float* x, const float* x_channel_shift, const float* w_rowmajor, const float* bias,
                    const float* out_relu_mask, float* y, int n, int cin, int64_t hw, int cout, int relu, int accumulate,
                    void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && w_rowmajor && y, MAUA_E_INVAL, "conv1x1_x3: null pointer");
    MAUA_REQUIRE(hw > 0 && conv_dims_ok(n, cin, 1, hw < (1 << 24) ? hw : 1, cout, 0), MAUA_E_INVAL, "conv1x1_x3: bad dims");
    MAUA_REQUIRE(hw < (1ll << 30) && (int64_t)cin * cout < (1ll << 30), MAUA_E_UNSUPPORTED, "conv1x1_x3: operand too large");
    ConvArgs a{};
    a.x = x;
    a.w = w_rowmajor;
    a.bias = bias;
    a.omask = out_relu_mask;
    a.y = y;
    a.Cin = cin;
    a.Cout = cout;
    a.H = 1;
    a.W = (int)hw;
    a.relu = relu;
    a.accumulate = accumulate;
    a.ws = (workspace && workspace_bytes >= maua_conv1x1_x3_workspace_bytes(n, cin, hw, cout)) ? (float*)workspace : nullptr;
    return conv1x1_x3_launch(a, x_channel_shift, n, (hipStream_t)stream);
}

}  // extern "C"
