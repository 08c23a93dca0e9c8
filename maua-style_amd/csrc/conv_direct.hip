// Direct (gather-form) convolution for geometry the MFMA path does not cover: strided filters such as NIN's
// 11x11 stride-4 stem (reference models.py:83).  One thread per output element; a wave covers 64 consecutive
// pixels of one channel, so the filter taps it needs are wave-uniform.  The forward reads the [tap][ci][co]
// bank, the backward-data reads the OIHW weights directly.
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

int conv_direct_fwd(const float* x, const float* mask, const float* wf, const float* bias, float* y, int n, int cin, int h,
                    int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int relu, int accumulate,
                    hipStream_t stream) {
    dim3 grid((unsigned)(((int64_t)oh * ow + 255) / 256), (unsigned)cout, (unsigned)n);
    hipLaunchKernelGGL(conv_direct_fwd_kernel, grid, dim3(256), 0, stream, x, mask, wf, bias, y, cin, h, w, cout, oh, ow,
                       kh, kw, stride, pad, relu, accumulate);
    return check_launch("conv_direct_fwd_kernel");
}

int conv_direct_bwd(const float* gy, const float* mask, const float* w_oihw, const float* omask, float* gx, int n, int cin,
                    int h, int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int accumulate,
                    hipStream_t stream) {
    dim3 grid((unsigned)(((int64_t)h * w + 255) / 256), (unsigned)cin, (unsigned)n);
    hipLaunchKernelGGL(conv_direct_bwd_kernel, grid, dim3(256), 0, stream, gy, mask, w_oihw, omask, gx, cin, h, w, cout,
                       oh, ow, kh, kw, stride, pad, accumulate);
    return check_launch("conv_direct_bwd_kernel");
}

}  // namespace maua
