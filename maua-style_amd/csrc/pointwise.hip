// Bandwidth-bound pieces of the hot path: ReLU, pooling, filter re-layout, MSE and TV losses with their
// gradients, small vector helpers.  All reductions are two-stage with a fixed summation order.
#include <stdarg.h>

#include "common.hpp"

namespace maua {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void finish_sum_kernel(const double* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    __shared__ double scratch[16];
    // each thread sums a fixed strided subset in order, then a fixed-shape tree: deterministic
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) out[0] = (float)(v * (double)scale);
}

// One workgroup per frame: every slot with a filled ledger record gets its loss (the record's partial sums added in
// finish_sum_kernel's order, times the record's scale) and the record is marked empty again; slots without a record keep the
// value an immediate-mode entry point left there.  Then total = sum of the slots, left to right (`total_loss += mod.loss`).
// The workgroup is four groups of 256 threads; group g takes the records g, g + 4, ...: each record is summed exactly as the one
// 256-thread workgroup of finish_sum_kernel sums it (thread t adds the partials t, t + 256, ... in order, then the fixed tree), four
// records at a time - the seven records of a VGG evaluation are two rounds of memory latency instead of seven (11.5 -> 6 us).
constexpr int LEDGER_GROUPS = 4;
__global__ void __launch_bounds__(256 * LEDGER_GROUPS)
loss_ledger_sum_kernel(double* __restrict__ ledger, int slots, float* __restrict__ losses, float* __restrict__ total,
                       double* __restrict__ exact) {
    __shared__ double scratch[LEDGER_GROUPS][4];
    __shared__ float kept[1 << 12];  // (maua_loss_ledger_sum: slots <= 4096)
    double* led = ledger + (int64_t)blockIdx.x * slots * LEDGER_STRIDE;
    float* out = losses + (int64_t)blockIdx.x * slots;
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    constexpr int PER = LEDGER_MAX / 256;  // partial sums per thread
    for (int s0 = 0; s0 < slots; s0 += LEDGER_GROUPS) {  // (uniform trip count: the barriers below are the whole workgroup's)
        const int s = s0 + grp;
        const bool have = s < slots;
        const double* rec = led + (int64_t)(have ? s : 0) * LEDGER_STRIDE;
        double part[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) part[i] = rec[2 + t + 256 * i];  // requested with the header, not behind it (the record is LEDGER_MAX long)
        const int n = have ? min((int)rec[0], LEDGER_MAX) : 0;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) v += t + 256 * i < n ? part[i] : 0.0;  // finish_sum_kernel's order (absent entries add +0)
        v = wave_sum(v);
        if (lane == 0) scratch[grp][wave] = v;
        __syncthreads();
        if (t == 0 && have) {
            if (n > 0) {
                double r = 0.0;
                for (int i = 0; i < 4; ++i) r += scratch[grp][i];
                const double scl = rec[1];
                kept[s] = (float)(r * scl);
                out[s] = kept[s];
                if (exact) exact[(int64_t)blockIdx.x * slots + s] = r * scl;  // (tests: the loss before its fp32 rounding)
                led[(int64_t)s * LEDGER_STRIDE] = 0.0;
            } else {
                kept[s] = out[s];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int s = 0; s < slots; ++s) tot += kept[s];
        total[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ void pack_filters_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int cout,
                                    int cin, int kh, int kw) {
    const int64_t total = (int64_t)cout * cin * kh * kw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int kx = (int)(r % kw);
        r /= kw;
        const int ky = (int)(r % kh);
        r /= kh;
        const int ci = (int)(r % cin);
        const int co = (int)(r / cin);
        const float v = w[e];
        if (wf) wf[((int64_t)(ky * kw + kx) * cin + ci) * cout + co] = v;
        if (wb) wb[((int64_t)((kh - 1 - ky) * kw + (kw - 1 - kx)) * cout + co) * cin + ci] = v;
    }
}

__global__ void relu_fwd_kernel(float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = x[i] > 0.f ? x[i] : 0.f;
}

__global__ void relu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                                int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        gx[i] = y[i] > 0.f ? gy[i] : 0.f;
}

__global__ void fill_kernel(float* __restrict__ x, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fmaf(a, x[i], y[i]);
}

__global__ void sum_small_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += in[i];  // same left-to-right order as `total_loss += mod.loss`
        out[0] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pooling.  Window of output (oy, ox): rows [oy*s, min(oy*s+k, H)), cols likewise (pad = 0 everywhere in the
// reference).  Max: first maximum in row-major scan order, NaN wins (ATen's max_pool2d rule).
__device__ __forceinline__ int window_argmax(const float* __restrict__ plane, int H, int W, int oy, int ox, int k, int s) {
    const int y0 = oy * s, x0 = ox * s;
    const int y1 = min(y0 + k, H), x1 = min(x0 + k, W);
    int best = y0 * W + x0;
    float bv = plane[best];
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
            const float v = plane[yy * W + xx];
            if (v > bv || v != v) {
                bv = v;
                best = yy * W + xx;
            }
        }
    return best;
}

__global__ void pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes, int H, int W, int OH,
                                int OW, int k, int s, int mode) {
    const int64_t total = planes * OH * OW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % OW);
        const int oy = (int)((e / OW) % OH);
        const int64_t pl = e / ((int64_t)OW * OH);
        const float* plane = x + pl * H * W;
        if (mode == 0) {
            y[e] = plane[window_argmax(plane, H, W, oy, ox, k, s)];
        } else {
            const int y0 = oy * s, x0 = ox * s, y1 = min(y0 + k, H), x1 = min(x0 + k, W);
            float acc = 0.f;
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = x0; xx < x1; ++xx) acc += plane[yy * W + xx];
            y[e] = acc / (float)((y1 - y0) * (x1 - x0));
        }
    }
}

__global__ void pool_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx,
                                int64_t planes, int H, int W, int OH, int OW, int k, int s, int mode, int relu_mask) {
    const int64_t total = planes * H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(e % W);
        const int iy = (int)((e / W) % H);
        const int64_t pl = e / ((int64_t)W * H);
        const float* plane = x + pl * H * W;
        const float* gplane = gy + pl * OH * OW;
        // output windows that contain (iy, ix)
        const int oy_lo = max(0, (iy - k + s) / s), oy_hi = min(OH - 1, iy / s);
        const int ox_lo = max(0, (ix - k + s) / s), ox_hi = min(OW - 1, ix / s);
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (oy * s > iy || oy * s + k <= iy || ox * s > ix || ox * s + k <= ix) continue;
                if (mode == 0) {
                    if (window_argmax(plane, H, W, oy, ox, k, s) == iy * W + ix) acc += gplane[oy * OW + ox];
                } else {
                    const int y0 = oy * s, x0 = ox * s, y1 = min(y0 + k, H), x1 = min(x0 + k, W);
                    acc += gplane[oy * OW + ox] / (float)((y1 - y0) * (x1 - x0));
                }
            }
        if (relu_mask && !(plane[iy * W + ix] > 0.f)) acc = 0.f;  // threshold_backward of the ReLU that produced x
        gx[e] = acc;
    }
}

// Max-pool backward for 3x3 windows at stride 2 (NIN's ceil-mode pools, reference models.py:77-80): gather form with
// the generic kernel's tie rule (first maximum in scan order, NaN wins), so it stays bit-reproducible - a scatter form
// would need float atomics over the overlapping windows.  A workgroup owns a 32 x 32 block of input pixels: it stages
// the 35 x 35 inputs under the 17 x 17 windows that touch the block in LDS, finds every window's arg-max ONCE (the
// per-pixel form re-scans up to four windows = 36 loads per pixel), then each pixel collects the gradient of the at
// most four windows that cover it and elected it, in ascending (oy, ox) order.  HBM traffic: x once (+ halo), gy once,
// gx once.
constexpr int P3_TO = 16, P3_TI = 2 * P3_TO + 3, P3_TW = P3_TO + 1;

__global__ void __launch_bounds__(256)
pool3s2_max_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, int H, int W, int OH,
                       int OW, int relu_mask) {
    __shared__ float xt[P3_TI * P3_TI];
    __shared__ int warg[P3_TW * P3_TW];
    __shared__ float wgrad[P3_TW * P3_TW];
    const int tid = threadIdx.x;
    const int oy0 = blockIdx.y * P3_TO, ox0 = blockIdx.x * P3_TO;
    const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;  // tile origin: two pixels of halo for the windows of row / column -1
    const float* plane = x + (int64_t)blockIdx.z * H * W;
    const float* gplane = gy + (int64_t)blockIdx.z * OH * OW;
    for (int i = tid; i < P3_TI * P3_TI; i += 256) {
        const int ty = i / P3_TI, tx = i - ty * P3_TI;
        const int Y = iy0 + ty, X = ix0 + tx;
        xt[i] = (Y >= 0 && Y < H && X >= 0 && X < W) ? plane[Y * W + X] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < P3_TW * P3_TW; i += 256) {
        const int wy = i / P3_TW, wx = i - wy * P3_TW;
        const int oy = oy0 - 1 + wy, ox = ox0 - 1 + wx;
        int arg = -1;
        float g = 0.f;
        if (oy >= 0 && oy < OH && ox >= 0 && ox < OW) {
            g = gplane[oy * OW + ox];
            int best = 2 * wy * P3_TI + 2 * wx;
            float bv = xt[best];
#pragma unroll
            for (int yy = 0; yy < 3; ++yy)
#pragma unroll
                for (int xx = 0; xx < 3; ++xx) {
                    if (2 * oy + yy >= H || 2 * ox + xx >= W) continue;
                    const int idx = (2 * wy + yy) * P3_TI + 2 * wx + xx;
                    const float v = xt[idx];
                    if (v > bv || v != v) {
                        bv = v;
                        best = idx;
                    }
                }
            arg = best;
        }
        warg[i] = arg;
        wgrad[i] = g;
    }
    __syncthreads();
    for (int i = tid; i < 4 * P3_TO * P3_TO; i += 256) {
        const int py = 2 + i / (2 * P3_TO), px = 2 + i % (2 * P3_TO);
        const int Y = iy0 + py, X = ix0 + px;
        if (Y >= H || X >= W) continue;
        const int self = py * P3_TI + px;
        float acc = 0.f;
        for (int wy = (py - 1) / 2; wy <= py / 2; ++wy)
            for (int wx = (px - 1) / 2; wx <= px / 2; ++wx)
                if (warg[wy * P3_TW + wx] == self) acc += wgrad[wy * P3_TW + wx];
        if (relu_mask && !(xt[self] > 0.f)) acc = 0.f;
        gx[(int64_t)blockIdx.z * H * W + Y * W + X] = acc;
    }
}

// 2x2 stride-2 max pooling on even-sized planes (every VGG pool): one thread per window, 8-byte accesses, no divisions.
// grid = (ceil(OW/256), OH, planes).  Same tie rule as window_argmax: scan order (0,0),(0,1),(1,0),(1,1), first max wins.
__global__ void __launch_bounds__(256)
pool2x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int W, int OW) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const int64_t OHl = gridDim.y;
    const float* p = x + ((int64_t)blockIdx.z * 2 * OHl + 2 * oy) * W + 2 * ox;
    const float2 r0 = *reinterpret_cast<const float2*>(p), r1 = *reinterpret_cast<const float2*>(p + W);
    float m = r0.x;
    if (r0.y > m || r0.y != r0.y) m = r0.y;
    if (r1.x > m || r1.x != r1.x) m = r1.x;
    if (r1.y > m || r1.y != r1.y) m = r1.y;
    y[((int64_t)blockIdx.z * OHl + oy) * OW + ox] = m;
}

__global__ void __launch_bounds__(256)
pool2x2_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, int W, int OW,
                   int relu_mask) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const int64_t OHl = gridDim.y;
    const int64_t base = ((int64_t)blockIdx.z * 2 * OHl + 2 * oy) * W + 2 * ox;
    const float2 r0 = *reinterpret_cast<const float2*>(x + base), r1 = *reinterpret_cast<const float2*>(x + base + W);
    float m = r0.x;
    int arg = 0;
    if (r0.y > m || r0.y != r0.y) { m = r0.y; arg = 1; }
    if (r1.x > m || r1.x != r1.x) { m = r1.x; arg = 2; }
    if (r1.y > m || r1.y != r1.y) { m = r1.y; arg = 3; }
    float g = gy[((int64_t)blockIdx.z * OHl + oy) * OW + ox];
    if (relu_mask && !(m > 0.f)) g = 0.f;  // the winner is a ReLU output: its threshold_backward
    *reinterpret_cast<float2*>(gx + base) = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(gx + base + W) = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
}

// The same pair with the decision kept: the forward pass leaves one byte per window - the winner's position in scan order
// (bits 1:0) and whether the winning value is <= 0 (bit 2: for a ReLU input, its threshold_backward) - so that the backward
// pass routes the gradient without reading the input map again (pool1 at 1024x1024: 603 -> 352 MB).  Same decisions, same bits.
// Layout of the bytes: [image][octet of channels][pooled pixel][channel within the octet] (C % 8 == 0) - the eight channels a staging
// item of conv_x3w.hip consumes are one 8-byte load there (maua_conv3x3_x3w_unpool), and its pooling epilogue stores four at a time.
__device__ __forceinline__ int64_t pool_code_index(int64_t image_channel, int64_t pooled_plane, int64_t pooled_pixel) {
    return ((image_channel >> 3) * pooled_plane + pooled_pixel) * 8 + (image_channel & 7);  // (C % 8 == 0: n * C + c keeps c's low bits)
}

// Planes of any extent >= 2 (round 4): `MaxPool2d(2, 2)` is floor mode (/root/reference/models.py:120), an odd plane's last row / column
// belongs to no window - it is not read on the way forward and gets a zero gradient on the way back.  ALIGNED (even width): the two
// elements of a window row are one 8-byte access.
template <bool ALIGNED>
__global__ void __launch_bounds__(256)
pool2x2_fwd_codes_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ codes, int H, int W, int OW) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const int64_t OHl = gridDim.y;
    const float* p = x + ((int64_t)blockIdx.z * H + 2 * oy) * W + 2 * ox;
    float2 r0, r1;
    if constexpr (ALIGNED) {
        r0 = *reinterpret_cast<const float2*>(p);
        r1 = *reinterpret_cast<const float2*>(p + W);
    } else {
        r0 = make_float2(p[0], p[1]);
        r1 = make_float2(p[W], p[W + 1]);
    }
    float m = r0.x;
    int arg = 0;
    if (r0.y > m || r0.y != r0.y) { m = r0.y; arg = 1; }
    if (r1.x > m || r1.x != r1.x) { m = r1.x; arg = 2; }
    if (r1.y > m || r1.y != r1.y) { m = r1.y; arg = 3; }
    const int64_t o = ((int64_t)blockIdx.z * OHl + oy) * OW + ox;
    y[o] = m;
    codes[pool_code_index(blockIdx.z, OHl * OW, oy * (int64_t)OW + ox)] = (unsigned char)(arg | (m > 0.f ? 0 : 4));
}

template <bool ALIGNED>
__global__ void __launch_bounds__(256)
pool2x2_bwd_codes_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ codes, float* __restrict__ gx, int H, int W,
                         int OW, int relu_mask) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const int64_t OHl = gridDim.y;
    const int64_t base = ((int64_t)blockIdx.z * H + 2 * oy) * W + 2 * ox;
    const int64_t o = ((int64_t)blockIdx.z * OHl + oy) * OW + ox;
    const int code = codes[pool_code_index(blockIdx.z, OHl * OW, oy * (int64_t)OW + ox)];
    const int arg = code & 3;
    float g = gy[o];
    if (relu_mask && (code & 4)) g = 0.f;
    if constexpr (ALIGNED) {
        *reinterpret_cast<float2*>(gx + base) = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f);
        *reinterpret_cast<float2*>(gx + base + W) = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
    } else {
        gx[base] = arg == 0 ? g : 0.f;
        gx[base + 1] = arg == 1 ? g : 0.f;
        gx[base + W] = arg == 2 ? g : 0.f;
        gx[base + W + 1] = arg == 3 ? g : 0.f;
        if (ox == OW - 1) gx[base + 2] = gx[base + W + 2] = 0.f;  // (odd width: the column no window owns)
    }
    if ((H & 1) && oy == (int)OHl - 1) {  // odd height: the row no window owns
        gx[base + 2 * W] = gx[base + 2 * W + 1] = 0.f;
        if ((W & 1) && ox == OW - 1) gx[base + 2 * W + 2] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// MSE: partial[b] = sum over the block's elements of (x-t)^2 (double); grad (+)= gs * (x - t).
__global__ void __launch_bounds__(256)
mse_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ grad, int64_t n, float gs,
           int accumulate, int mask_by_x, double* __restrict__ partial, double* __restrict__ rec, float rec_scale) {
    __shared__ double scratch[16];
    if (rec && blockIdx.x == 0 && threadIdx.x == 0) {  // ledger record header: number of partial sums, scale
        rec[0] = (double)gridDim.x;
        rec[1] = (double)rec_scale;
    }
    double acc = 0.0;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto one = [&](int64_t e, float xv, float tv, float gv) {
        const float d = xv - tv;
        acc += (double)d * (double)d;
        if (grad) {
            float g = accumulate ? fmaf(gs, d, gv) : gs * d;
            if (mask_by_x && !(xv > 0.f)) g = 0.f;  // x is a ReLU output: apply its threshold_backward here
            grad[e] = g;
        }
    };
    for (; i + 3 * step < n; i += 4 * step) {  // four elements' loads in flight; the additions in the element order of the plain loop
        float xv[4], tv[4], gv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xv[k] = x[i + k * step];
            tv[k] = t[i + k * step];
            if (grad && accumulate) gv[k] = grad[i + k * step];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) one(i + k * step, xv[k], tv[k], gv[k]);
    }
    for (; i < n; i += step) one(i, x[i], t[i], grad && accumulate ? grad[i] : 0.f);
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// Weighted MSE of the temporal ContentLoss (reference loss.py:52-56: the reliability mask multiplies the INPUT only):
// partial[b] = sum (x*w - t)^2 ; grad (+)= gs * w * (x*w - t).  w has `wplanes` planes of `plane` elements (1 = one mask
// broadcast over the channels, or as many as x).
__global__ void __launch_bounds__(256)
mse_weighted_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ t,
                    float* __restrict__ grad, int64_t n, int64_t plane, int wplanes, float gs, int accumulate,
                    double* __restrict__ partial) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pl = i / plane;
        const float wv = w[(pl % wplanes) * plane + (i - pl * plane)];
        const float d = x[i] * wv - t[i];
        acc += (double)d * (double)d;
        if (grad) grad[i] = accumulate ? fmaf(gs * wv, d, grad[i]) : gs * wv * d;
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// TV: loss = strength * (sum |x[y+1]-x[y]| + sum |x[x+1]-x[x]|); d/dx via sign() of the four neighbours' differences.
__global__ void __launch_bounds__(256)
tv_kernel(const float* __restrict__ x, float* __restrict__ grad, int64_t planes, int H, int W, float strength,
          int accumulate, double* __restrict__ partial, double* __restrict__ rec) {
    __shared__ double scratch[16];
    if (rec && blockIdx.x == 0 && threadIdx.x == 0) {
        rec[0] = (double)gridDim.x;
        rec[1] = (double)strength;
    }
    const int64_t total = planes * H * W;
    double acc = 0.0;
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    auto one = [&](int64_t e, int ix, int iy) {
        const float c = x[e];
        const float dn = iy + 1 < H ? x[e + W] : c, up = iy > 0 ? x[e - W] : c, rt = ix + 1 < W ? x[e + 1] : c, lf = ix > 0 ? x[e - 1] : c;
        float g = 0.f;
        if (iy + 1 < H) {
            const float d = dn - c;  // this element is the "upper" one of the pair: d/dc = -sign(d)
            acc += fabs((double)d);
            g -= sgn(d);
        }
        if (iy > 0) g += sgn(c - up);
        if (ix + 1 < W) {
            const float d = rt - c;
            acc += fabs((double)d);
            g -= sgn(d);
        }
        if (ix > 0) g += sgn(c - lf);
        if (grad) grad[e] = accumulate ? fmaf(strength, g, grad[e]) : strength * g;
    };
    if (total < (1ll << 31)) {  // 32-bit index arithmetic (a 64-bit division costs more than the element's five loads)
        const unsigned tot = (unsigned)total, step = gridDim.x * blockDim.x, Wu = (unsigned)W, Hu = (unsigned)H;
        for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += step) {
            const unsigned row = e / Wu;
            one((int64_t)e, (int)(e - row * Wu), (int)(row % Hu));
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
            one(e, (int)(e % W), (int)((e / W) % H));
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// Adam, single-tensor form of torch.optim.Adam: exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; x -= lr/(1-b1^t) * m/denom.
__global__ void adam_kernel(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                            float sqrt_bc2, float step_size, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + one_minus_b1 * (gi - m[i]);
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        x[i] = x[i] + (-step_size * mi) / denom;  // addcdiv_(exp_avg, denom, value=-step_size)
    }
}

static inline int ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace maua

using namespace maua;

// Depth to space: in[n][(ry r + rx) c_out + c][qy][qx] -> out[n][c][r qy + ry][r qx + rx] for the H x W pixels that exist (sites beyond the
// input's extent: zero).  The tail of the strided layers' backward pass as a stride-1 convolution over the output sites (maua_depth_to_space).
__global__ void __launch_bounds__(256)
depth_to_space_kernel(const float* __restrict__ in, float* __restrict__ out, int c_out, int r, int QH, int QW, int H, int W, int accumulate) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, nc = blockIdx.z;
    if (x >= W) return;
    const int n = nc / c_out, c = nc - n * c_out;
    const int qy = y / r, qx = x / r, ry = y - qy * r, rx = x - qx * r;
    float v = 0.f;
    if (qy < QH && qx < QW) v = in[(((int64_t)n * r * r + ry * r + rx) * c_out + c) * ((int64_t)QH * QW) + (int64_t)qy * QW + qx];
    float* o = out + ((int64_t)nc * H + y) * W + x;
    *o = accumulate ? *o + v : v;
}

// Space to depth: out[n][(ry r + rx) c_in + c][qy][qx] = in[n][c][r qy + ry][r qx + rx] (0 beyond the h x w pixels): the head of a strided
// layer's forward pass as a stride-1 convolution over sites (maua_space_to_depth).
__global__ void __launch_bounds__(256)
space_to_depth_kernel(const float* __restrict__ in, float* __restrict__ out, int c_in, int r, int QH, int QW, int H, int W) {
    const int qx = blockIdx.x * 256 + threadIdx.x, qy = blockIdx.y, z = blockIdx.z;  // z = (n, ry, rx, c)
    if (qx >= QW) return;
    const int c = z % c_in, ph = (z / c_in) % (r * r), n = z / (c_in * r * r);
    const int y = r * qy + ph / r, x = r * qx + ph % r;
    float v = 0.f;
    if (y < H && x < W) v = in[(((int64_t)n * c_in + c) * H + y) * W + x];
    out[((int64_t)z * QH + qy) * QW + qx] = v;
}

extern "C" {

int maua_abi_version(void) { return 2; }
const char* maua_last_error(void) { return maua::g_err; }

int maua_conv_pack_filters(const float* w, float* wf, float* wb, int cout, int cin, int kh, int kw, maua_stream_t stream) {
    MAUA_REQUIRE(w && (wf || wb) && cout > 0 && cin > 0 && kh > 0 && kw > 0, MAUA_E_INVAL, "conv_pack_filters: bad args");
    const int64_t total = (int64_t)cout * cin * kh * kw;
    hipLaunchKernelGGL(pack_filters_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, wf, wb, cout, cin,
                       kh, kw);
    return check_launch("pack_filters_kernel");
}

int maua_relu_fwd(float* x, int64_t count, maua_stream_t stream) {
    MAUA_REQUIRE(x && count > 0, MAUA_E_INVAL, "relu_fwd: bad args");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, x, count);
    return check_launch("relu_fwd_kernel");
}

int maua_relu_bwd(const float* gy, const float* y, float* gx, int64_t count, maua_stream_t stream) {
    MAUA_REQUIRE(gy && y && gx && count > 0, MAUA_E_INVAL, "relu_bwd: bad args");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, gy, y, gx, count);
    return check_launch("relu_bwd_kernel");
}

int maua_pool_out_size(int in, int k, int stride, int ceil_mode) {
    // ATen pooling_output_shape with pad 0, dilation 1: floor/ceil((in - k) / stride) + 1, and in ceil mode the last
    // window must start inside the input.
    if (in <= 0 || k <= 0 || stride <= 0 || in < k) return 0;
    int o = (in - k + (ceil_mode ? stride - 1 : 0)) / stride + 1;
    if (ceil_mode && (o - 1) * stride >= in) --o;
    return o;
}

int maua_pool2d_fwd(const float* x, float* y, int n, int c, int h, int w, int k, int stride, int ceil_mode, int mode,
                    maua_stream_t stream) {
    MAUA_REQUIRE(x && y && conv_dims_ok(n, c, h, w, 1, 0) && k > 0 && k <= 64 && stride > 0 && stride <= 64 && (mode == 0 || mode == 1),
                 MAUA_E_INVAL, "pool2d_fwd: bad args");
    const int oh = maua_pool_out_size(h, k, stride, ceil_mode), ow = maua_pool_out_size(w, k, stride, ceil_mode);
    MAUA_REQUIRE(oh > 0 && ow > 0, MAUA_E_UNSUPPORTED, "pool2d_fwd: input %dx%d smaller than window %d", h, w, k);
    if (mode == 0 && k == 2 && stride == 2 && h % 2 == 0 && w % 2 == 0 && (int64_t)n * c <= 65535 && oh <= 65535) {
        hipLaunchKernelGGL(pool2x2_fwd_kernel, dim3((ow + 255) / 256, oh, n * c), dim3(256), 0, (hipStream_t)stream, x, y, w, ow);
        return check_launch("pool2x2_fwd_kernel");
    }
    const int64_t total = (int64_t)n * c * oh * ow;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, (int64_t)n * c, h, w,
                       oh, ow, k, stride, mode);
    return check_launch("pool_fwd_kernel");
}

int maua_pool2d_bwd(const float* gy, const float* x, float* gx, int n, int c, int h, int w, int k, int stride,
                    int ceil_mode, int mode, int relu_mask_by_x, maua_stream_t stream) {
    MAUA_REQUIRE(gy && x && gx && conv_dims_ok(n, c, h, w, 1, 0) && k > 0 && k <= 64 && stride > 0 && stride <= 64 && (mode == 0 || mode == 1),
                 MAUA_E_INVAL, "pool2d_bwd: bad args");
    const int oh = maua_pool_out_size(h, k, stride, ceil_mode), ow = maua_pool_out_size(w, k, stride, ceil_mode);
    MAUA_REQUIRE(oh > 0 && ow > 0, MAUA_E_UNSUPPORTED, "pool2d_bwd: input %dx%d smaller than window %d", h, w, k);
    if (mode == 0 && k == 2 && stride == 2 && h % 2 == 0 && w % 2 == 0 && (int64_t)n * c <= 65535 && oh <= 65535) {
        hipLaunchKernelGGL(pool2x2_bwd_kernel, dim3((ow + 255) / 256, oh, n * c), dim3(256), 0, (hipStream_t)stream, gy, x, gx, w,
                           ow, relu_mask_by_x);
        return check_launch("pool2x2_bwd_kernel");
    }
    if (mode == 0 && k == 3 && stride == 2 && (int64_t)n * c <= 65535 && h <= 65535 * 32) {
        hipLaunchKernelGGL(pool3s2_max_bwd_kernel, dim3((w + 2 * P3_TO - 1) / (2 * P3_TO), (h + 2 * P3_TO - 1) / (2 * P3_TO), n * c),
                           dim3(256), 0, (hipStream_t)stream, gy, x, gx, h, w, oh, ow, relu_mask_by_x);
        return check_launch("pool3s2_max_bwd_kernel");
    }
    const int64_t total = (int64_t)n * c * h * w;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, gy, x, gx, (int64_t)n * c,
                       h, w, oh, ow, k, stride, mode, relu_mask_by_x);
    return check_launch("pool_bwd_kernel");
}

int maua_pool2x2_codes_supported(int n, int c, int h, int w) {
    return conv_dims_ok(n, c, h, w, 1, 0) && h >= 2 && w >= 2 && c % 8 == 0 && (int64_t)n * c <= 65535 && h / 2 <= 65535;
}

int maua_pool2x2_fwd_codes(const float* x, float* y, unsigned char* codes, int n, int c, int h, int w, maua_stream_t stream) {
    MAUA_REQUIRE(x && y && codes, MAUA_E_INVAL, "pool2x2_fwd_codes: null pointer");
    MAUA_REQUIRE(maua_pool2x2_codes_supported(n, c, h, w), MAUA_E_UNSUPPORTED, "pool2x2_fwd_codes: needs planes of 2 x 2 and more, c %% 8 == 0");
    if (w % 2 == 0)
        hipLaunchKernelGGL(pool2x2_fwd_codes_kernel<true>, dim3((w / 2 + 255) / 256, h / 2, n * c), dim3(256), 0, (hipStream_t)stream, x, y,
                           codes, h, w, w / 2);
    else
        hipLaunchKernelGGL(pool2x2_fwd_codes_kernel<false>, dim3((w / 2 + 255) / 256, h / 2, n * c), dim3(256), 0, (hipStream_t)stream, x, y,
                           codes, h, w, w / 2);
    return check_launch("pool2x2_fwd_codes_kernel");
}

int maua_pool2x2_bwd_codes(const float* gy, const unsigned char* codes, float* gx, int n, int c, int h, int w, int relu_mask,
                           maua_stream_t stream) {
    MAUA_REQUIRE(gy && gx && codes, MAUA_E_INVAL, "pool2x2_bwd_codes: null pointer");
    MAUA_REQUIRE(maua_pool2x2_codes_supported(n, c, h, w), MAUA_E_UNSUPPORTED, "pool2x2_bwd_codes: needs planes of 2 x 2 and more, c %% 8 == 0");
    if (w % 2 == 0)
        hipLaunchKernelGGL(pool2x2_bwd_codes_kernel<true>, dim3((w / 2 + 255) / 256, h / 2, n * c), dim3(256), 0, (hipStream_t)stream, gy,
                           codes, gx, h, w, w / 2, relu_mask);
    else
        hipLaunchKernelGGL(pool2x2_bwd_codes_kernel<false>, dim3((w / 2 + 255) / 256, h / 2, n * c), dim3(256), 0, (hipStream_t)stream, gy,
                           codes, gx, h, w, w / 2, relu_mask);
    return check_launch("pool2x2_bwd_codes_kernel");
}

size_t maua_reduce_workspace_bytes(int64_t count) { return (size_t)reduce_blocks(count, 1024) * sizeof(double); }

int maua_mse_fwd_bwd(const float* x, const float* target, float* grad, int64_t count, float loss_scale, float grad_scale,
                     int accumulate, int mask_grad_by_x, float* loss_out, void* workspace, size_t workspace_bytes,
                     maua_stream_t stream) {
    MAUA_REQUIRE(x && target && loss_out && workspace && count > 0, MAUA_E_INVAL, "mse_fwd_bwd: bad args");
    const int nb = reduce_blocks(count, 1024);
    MAUA_REQUIRE(workspace_bytes >= nb * sizeof(double), MAUA_E_WORKSPACE, "mse_fwd_bwd: workspace %zu < %zu",
                 workspace_bytes, nb * sizeof(double));
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, target, grad, count, grad_scale,
                       accumulate, mask_grad_by_x, (double*)workspace, (double*)nullptr, 0.f);
    int rc = check_launch("mse_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nb,
                       loss_scale, loss_out);
    return check_launch("finish_sum_kernel");
}

int maua_mse_weighted_fwd_bwd(const float* x, const float* weights, const float* target, float* grad, int64_t planes,
                              int64_t plane, int weight_planes, float loss_scale, float grad_scale, int accumulate,
                              float* loss_out, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && weights && target && loss_out && workspace && planes > 0 && plane > 0, MAUA_E_INVAL,
                 "mse_weighted_fwd_bwd: bad args");
    MAUA_REQUIRE(weight_planes == 1 || weight_planes == planes, MAUA_E_INVAL,
                 "mse_weighted_fwd_bwd: weights need 1 or %lld planes, got %d", (long long)planes, weight_planes);
    const int64_t count = planes * plane;
    const int nb = reduce_blocks(count, 1024);
    MAUA_REQUIRE(workspace_bytes >= nb * sizeof(double), MAUA_E_WORKSPACE, "mse_weighted_fwd_bwd: workspace too small");
    hipLaunchKernelGGL(mse_weighted_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, weights, target, grad, count, plane,
                       weight_planes, grad_scale, accumulate, (double*)workspace);
    int rc = check_launch("mse_weighted_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nb,
                       loss_scale, loss_out);
    return check_launch("finish_sum_kernel");
}

int maua_tv_fwd_bwd(const float* x, float* grad, int n, int c, int h, int w, float strength, int accumulate,
                    float* loss_out, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && loss_out && workspace && conv_dims_ok(n, c, h, w, 1, 0), MAUA_E_INVAL, "tv_fwd_bwd: bad args");
    const int64_t count = (int64_t)n * c * h * w;
    const int nb = reduce_blocks(count, 1024);
    MAUA_REQUIRE(workspace_bytes >= nb * sizeof(double), MAUA_E_WORKSPACE, "tv_fwd_bwd: workspace too small");
    hipLaunchKernelGGL(tv_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, grad, (int64_t)n * c, h, w, strength,
                       accumulate, (double*)workspace, (double*)nullptr);
    int rc = check_launch("tv_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nb, strength,
                       loss_out);
    return check_launch("finish_sum_kernel");
}

size_t maua_loss_ledger_bytes(int frames, int slots) {
    if (frames <= 0 || slots <= 0 || frames > (1 << 16) || slots > (1 << 12)) return 0;
    return (size_t)frames * slots * LEDGER_STRIDE * sizeof(double);
}

int maua_mse_fwd_bwd_ledger(const float* x, const float* target, float* grad, int64_t count, float loss_scale, float grad_scale,
                            int accumulate, int mask_grad_by_x, double* ledger, int slot, maua_stream_t stream) {
    MAUA_REQUIRE(x && target && ledger && slot >= 0 && slot < (1 << 28) && count > 0, MAUA_E_INVAL, "mse_fwd_bwd_ledger: bad args");
    const int nb = reduce_blocks(count, 1024);
    double* rec = ledger + (int64_t)slot * LEDGER_STRIDE;
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, target, grad, count, grad_scale, accumulate,
                       mask_grad_by_x, rec + 2, rec, loss_scale);
    return check_launch("mse_kernel");
}

int maua_tv_fwd_bwd_ledger(const float* x, float* grad, int n, int c, int h, int w, float strength, int accumulate,
                           double* ledger, int slot, maua_stream_t stream) {
    MAUA_REQUIRE(x && ledger && slot >= 0 && slot < (1 << 28) && conv_dims_ok(n, c, h, w, 1, 0), MAUA_E_INVAL, "tv_fwd_bwd_ledger: bad args");
    const int64_t count = (int64_t)n * c * h * w;
    const int nb = reduce_blocks(count, 1024);
    double* rec = ledger + (int64_t)slot * LEDGER_STRIDE;
    hipLaunchKernelGGL(tv_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, grad, (int64_t)n * c, h, w, strength, accumulate,
                       rec + 2, rec);
    return check_launch("tv_kernel");
}

int maua_loss_ledger_sum(double* ledger, int frames, int slots, float* losses, float* totals, maua_stream_t stream) {
    MAUA_REQUIRE(ledger && losses && totals && frames > 0 && frames <= (1 << 16) && slots > 0 && slots <= (1 << 12), MAUA_E_INVAL,
                 "loss_ledger_sum: bad args");
    hipLaunchKernelGGL(loss_ledger_sum_kernel, dim3(frames), dim3(256 * LEDGER_GROUPS), 0, (hipStream_t)stream, ledger, slots, losses, totals,
                       (double*)nullptr);
    return check_launch("loss_ledger_sum_kernel");
}

int maua_loss_ledger_sum_f64(double* ledger, int frames, int slots, float* losses, float* totals, double* losses_f64,
                             maua_stream_t stream) {
    MAUA_REQUIRE(ledger && losses && totals && losses_f64 && frames > 0 && frames <= (1 << 16) && slots > 0 && slots <= (1 << 12),
                 MAUA_E_INVAL, "loss_ledger_sum_f64: bad args");
    hipLaunchKernelGGL(loss_ledger_sum_kernel, dim3(frames), dim3(256 * LEDGER_GROUPS), 0, (hipStream_t)stream, ledger, slots, losses, totals,
                       losses_f64);
    return check_launch("loss_ledger_sum_kernel");
}

int maua_depth_to_space(const float* in, float* out, int n, int c_out, int r, int qh, int qw, int h, int w, int accumulate, maua_stream_t stream) {
    MAUA_REQUIRE(in && out && n > 0 && c_out > 0 && r > 0 && r <= 16 && qh > 0 && qw > 0 && h > 0 && w > 0 && (int64_t)n * c_out <= 65535 && h <= 65535,
                 MAUA_E_INVAL, "depth_to_space: bad args");
    hipLaunchKernelGGL(depth_to_space_kernel, dim3((w + 255) / 256, h, n * c_out), dim3(256), 0, (hipStream_t)stream, in, out, c_out, r, qh, qw, h, w, accumulate);
    return check_launch("depth_to_space_kernel");
}

int maua_space_to_depth(const float* in, float* out, int n, int c_in, int r, int h, int w, int qh, int qw, maua_stream_t stream) {
    MAUA_REQUIRE(in && out && n > 0 && c_in > 0 && r > 0 && r <= 16 && qh > 0 && qw > 0 && h > 0 && w > 0 && (int64_t)n * c_in * r * r <= 65535 && qh <= 65535,
                 MAUA_E_INVAL, "space_to_depth: bad args");
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((qw + 255) / 256, qh, n * c_in * r * r), dim3(256), 0, (hipStream_t)stream, in, out, c_in, r, qh, qw, h, w);
    return check_launch("space_to_depth_kernel");
}

int maua_fill(float* x, int64_t count, float value, maua_stream_t stream) {
    MAUA_REQUIRE(x && count > 0, MAUA_E_INVAL, "fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, x, count, value);
    return check_launch("fill_kernel");
}

int maua_axpy(float* y, const float* x, float alpha, int64_t count, maua_stream_t stream) {
    MAUA_REQUIRE(x && y && count > 0, MAUA_E_INVAL, "axpy: bad args");
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, count);
    return check_launch("axpy_kernel");
}

int maua_sum_small(const float* in, int count, float* out, maua_stream_t stream) {
    MAUA_REQUIRE(in && out && count > 0, MAUA_E_INVAL, "sum_small: bad args");
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, count, out);
    return check_launch("sum_small_kernel");
}

int maua_adam_step(float* x, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, int step, float lr,
                   float beta1, float beta2, float eps, maua_stream_t stream) {
    MAUA_REQUIRE(x && grad && exp_avg && exp_avg_sq && count > 0 && step >= 1, MAUA_E_INVAL, "adam_step: bad args");
    // scalar prep in double on the host like torch's python-float arithmetic (bias corrections are python floats)
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, x, grad, exp_avg, exp_avg_sq,
                       count, (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), sqrt_bc2, step_size,
                       eps);
    return check_launch("adam_kernel");
}

}  // extern "C"
