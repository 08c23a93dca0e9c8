// Image-space steps that sit between two runs of the optimisation loop, kept on the device so
// the pastiche never leaves HBM between scales:
//   * utils.match_histogram (reference utils.py:88-151): PCA colour transfer = per-channel mean + 3x3 covariance of the
//     jittered image (one reduction pass, fp64 partials, fixed order), a 3x3 symmetric eigen-problem (one thread, fp64
//     Jacobi; matrix square roots are unique, so the result does not depend on the eigen-solver), and an affine 3x3 colour
//     map (one elementwise pass);
//   * F.interpolate(mode="bilinear", align_corners=False) as style.img_img calls it (style.py:38-66);
//   * load.deprocess (load.py:47-52): mean back on, BGR -> RGB, /255, clamp, x255, truncation to 8 bits, HWC.
// All HBM-bound on 3-channel images (12.6 MB at 1024x1024): a few microseconds each.
#include "common.hpp"

namespace maua {

// Layout.  The reference views the (B, C, H, W) tensor as (B, W, H, C) and, for the whole batch at once, reshapes the centred
// tensor `h.permute(0, 3, 1, 2).reshape(C, -1)` (utils.py:91): the B*C planes (order p = 3 b + c) are cut into C = 3 rows of B
// whole planes each, so "component" k of pseudo-pixel (j, h, w) is plane k * B + j - for a single image (B = 1) simply
// channel k, for a clip a mix of frames and channels (that is what the reference computes, and the colour map is applied
// through the same reshape, so it is reproduced, not repaired).  Means are per TRUE channel c = p % 3.
// Jitter: `1e-3 * randn(size=frame.shape)` is drawn in the (B, W, H, C) view, so element (b, c, h, w) meets
// noise[((b * W + w) * H + h) * 3 + c].  Lanes run along h: the noise values of a lane are contiguous and a wave reads 768
// contiguous bytes; the image reads are strided (one row per lane) but a workgroup walks 16 neighbouring columns, so every
// 64-byte sector it touches is used completely out of L1.
constexpr int ST_ROWS = 64, ST_COLS = 16;

__device__ __forceinline__ void jittered_pixel(const float* __restrict__ x, const float* __restrict__ noise, float amp, int B, int j,
                                               int H, int W, int h, int w, float (&v)[3]) {
    const int64_t plane = (int64_t)H * W;
    const int64_t o = (int64_t)h * W + w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = k * B + j, b = p / 3, c = p - 3 * b;
        const float xv = x[p * plane + o];
        // frame + 1e-3 * randn, both roundings as torch performs them (fp32 multiply, then fp32 add)
        v[k] = noise ? xv + __fmul_rn(amp, noise[(((int64_t)b * W + w) * H + h) * 3 + c]) : xv;
    }
}

// partial[block][9] = sum x_c (3) and sum x_a x_b for (a, b) in (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) over the block's tile
__global__ void __launch_bounds__(256) channel_stats_kernel(const float* __restrict__ x, const float* __restrict__ noise, float amp,
                                                            int B, int j, int H, int W, int tiles_w, double* __restrict__ partial) {
    __shared__ double scratch[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h = th * ST_ROWS + lane;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (h < H) {
#pragma unroll
        for (int k = 0; k < ST_COLS / 4; ++k) {
            const int w = tw * ST_COLS + wave * (ST_COLS / 4) + k;
            if (w < W) {
                float v[3];
                jittered_pixel(x, noise, amp, B, j, H, W, h, w, v);
                const double a = v[0], b = v[1], c = v[2];
                s[0] += a;
                s[1] += b;
                s[2] += c;
                s[3] += a * a;
                s[4] += a * b;
                s[5] += a * c;
                s[6] += b * b;
                s[7] += b * c;
                s[8] += c * c;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double t = block_sum(s[i], scratch);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 9 + i] = t;
    }
}

// stats[9] = fixed-order sums of the partials (same layout)
__global__ void __launch_bounds__(256) channel_stats_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ stats) {
    __shared__ double scratch[16];
    for (int i = 0; i < 9; ++i) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += partial[(int64_t)b * 9 + i];
        v = block_sum(v, scratch);
        if (threadIdx.x == 0) stats[i] = v;
    }
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (fp64): a = v diag(e) v^T.
__device__ void jacobi3(double a[3][3], double v[3][3], double e[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) e[i] = a[i][i];
}

// Mean per true channel and covariance (+ eps I) of the 3 reshaped rows from the raw sums st[B][9] (one 9-tuple per plane
// slot j, `hw` pixels each); Q = cov^(1/2) (negative eigenvalues -> 0, utils.py:129 `Et[Et != Et] = 0`).
__device__ bool sqrt_cov(const double* __restrict__ st, int B, double hw, double eps, double mu[3], double q[3][3]) {
    double a[3][3], v[3][3], e[3];
    const double count = hw * B;
    for (int c = 0; c < 3; ++c) mu[c] = 0.0;
    for (int j = 0; j < B; ++j)
        for (int k = 0; k < 3; ++k) mu[(k * B + j) % 3] += st[j * 9 + k];
    for (int c = 0; c < 3; ++c) mu[c] /= count;
    const int idx[3][3] = {{3, 4, 5}, {4, 6, 7}, {5, 7, 8}};
    for (int r1 = 0; r1 < 3; ++r1)
        for (int r2 = 0; r2 < 3; ++r2) {
            double acc = 0.0;
            for (int j = 0; j < B; ++j) {  // sum (x1 - m1)(x2 - m2) = sum x1 x2 - m1 sum x2 - m2 sum x1 + hw m1 m2
                const double m1 = mu[(r1 * B + j) % 3], m2 = mu[(r2 * B + j) % 3];
                acc += st[j * 9 + idx[r1][r2]] - m1 * st[j * 9 + r2] - m2 * st[j * 9 + r1] + hw * m1 * m2;
            }
            a[r1][r2] = acc / count + (r1 == r2 ? eps : 0.0);
        }
    bool ok = true;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) ok = ok && isfinite(a[i][j]);
    if (!ok) return false;
    jacobi3(a, v, e);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += v[i][k] * (e[k] > 0.0 ? sqrt(e[k]) : 0.0) * v[j][k];
            q[i][j] = acc;
        }
    return true;
}

// coef[16]: M = Qs Qt^-1 (9, row-major), mu_t (3), mu_s (3), ok flag (1 / 0).  A singular Qt or non-finite statistics are
// where the reference's torch.inverse / symeig raise and its `except RuntimeError` returns the untouched image: flag 0.
__global__ void color_match_solve_kernel(const double* __restrict__ st_t, int B, double hw_t, const double* __restrict__ st_s,
                                         double hw_s, double eps, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double mu_t[3], mu_s[3], qt[3][3], qs[3][3];
    bool ok = sqrt_cov(st_t, B, hw_t, eps, mu_t, qt) && sqrt_cov(st_s, 1, hw_s, eps, mu_s, qs);
    double inv[3][3];
    if (ok) {
        const double c00 = qt[1][1] * qt[2][2] - qt[1][2] * qt[2][1], c01 = qt[1][2] * qt[2][0] - qt[1][0] * qt[2][2],
                     c02 = qt[1][0] * qt[2][1] - qt[1][1] * qt[2][0];
        const double det = qt[0][0] * c00 + qt[0][1] * c01 + qt[0][2] * c02;
        ok = isfinite(det) && det != 0.0;
        if (ok) {
            const double r = 1.0 / det;
            inv[0][0] = c00 * r;
            inv[1][0] = c01 * r;
            inv[2][0] = c02 * r;
            inv[0][1] = (qt[0][2] * qt[2][1] - qt[0][1] * qt[2][2]) * r;
            inv[1][1] = (qt[0][0] * qt[2][2] - qt[0][2] * qt[2][0]) * r;
            inv[2][1] = (qt[0][1] * qt[2][0] - qt[0][0] * qt[2][1]) * r;
            inv[0][2] = (qt[0][1] * qt[1][2] - qt[0][2] * qt[1][1]) * r;
            inv[1][2] = (qt[0][2] * qt[1][0] - qt[0][0] * qt[1][2]) * r;
            inv[2][2] = (qt[0][0] * qt[1][1] - qt[0][1] * qt[1][0]) * r;
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            if (ok)
                for (int k = 0; k < 3; ++k) acc += qs[i][k] * inv[k][j];
            ok = ok && isfinite(acc);
            coef[i * 3 + j] = (float)acc;
        }
    for (int c = 0; c < 3; ++c) {
        coef[9 + c] = ok ? (float)mu_t[c] : 0.f;
        coef[12 + c] = ok ? (float)mu_s[c] : 0.f;
    }
    coef[15] = ok ? 1.f : 0.f;
}

// out (+)= weight * (M (x_n - mu_t) + mu_s); when ANY of the `n_coef` solves of this call failed, out = x (the reference's
// `except RuntimeError: return backup` covers the whole call).  `coef` = this source's 16 floats, `all_coef` = every source's.
__global__ void __launch_bounds__(256) color_match_apply_kernel(const float* __restrict__ x, const float* __restrict__ noise, float amp,
                                                                const float* __restrict__ coef, const float* __restrict__ all_coef,
                                                                int n_coef, float weight, int accumulate, int B, int j, int H, int W,
                                                                int tiles_w, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h = th * ST_ROWS + lane;
    if (h >= H) return;
    bool ok = true;
    for (int i = 0; i < n_coef; ++i) ok = ok && all_coef[i * 16 + 15] > 0.f;
    float m[9], mt[3], ms[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = coef[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mt[c] = coef[9 + c];
        ms[c] = coef[12 + c];
    }
    const int64_t plane = (int64_t)H * W;
    int pl[3];  // plane of component k, and its true channel (for the two means)
    float mtk[3], msk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pl[k] = k * B + j;
        mtk[k] = mt[pl[k] % 3];
        msk[k] = ms[pl[k] % 3];
    }
#pragma unroll
    for (int k = 0; k < ST_COLS / 4; ++k) {
        const int w = tw * ST_COLS + wave * (ST_COLS / 4) + k;
        if (w >= W) continue;
        const int64_t o = (int64_t)h * W + w;
        if (!ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out[pl[c] * plane + o] = x[pl[c] * plane + o];
            continue;
        }
        float v[3];
        jittered_pixel(x, noise, amp, B, j, H, W, h, w, v);
        const float d0 = v[0] - mtk[0], d1 = v[1] - mtk[1], d2 = v[2] - mtk[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r = fmaf(m[c * 3 + 2], d2, fmaf(m[c * 3 + 1], d1, m[c * 3] * d0)) + msk[c];
            r *= weight;  // `matched / len(sources)` (utils.py:146)
            out[pl[c] * plane + o] = accumulate ? out[pl[c] * plane + o] + r : r;
        }
    }
}

// ATen upsample_bilinear2d, align_corners=False, antialias=False: src = max(0, scale * (dst + 0.5) - 0.5) in fp32
// (area_pixel_compute_source_index), neighbours clamped at the far edge, weights (1 - lambda, lambda).
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H,
                                                              int W, int OH, int OW, float scale_h, float scale_w) {
    const int64_t total = (int64_t)planes * OH * OW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % OW);
        const int oy = (int)((e / OW) % OH);
        const int64_t pl = e / ((int64_t)OW * OH);
        float sy = scale_h * ((float)oy + 0.5f) - 0.5f, sx = scale_w * ((float)ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p = x + pl * (int64_t)H * W;
        y[e] = hy * (hx * p[(int64_t)y0 * W + x0] + lx * p[(int64_t)y0 * W + x1]) +
               ly * (hx * p[(int64_t)y1 * W + x0] + lx * p[(int64_t)y1 * W + x1]);
    }
}

// load.deprocess (reference load.py:47-52) + ToPILImage's truncation: out[h][w][rgb] = byte(clamp((x[bgr] + mean) / 255, 0, 1) * 255)
__global__ void __launch_bounds__(256) deprocess_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int H, int W,
                                                           float mean_b, float mean_g, float mean_r) {
    const int64_t total = (int64_t)H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const float mean[3] = {mean_b, mean_g, mean_r};
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // c = channel of the BGR input; RGB position 2 - c
            float v = __fdiv_rn(x[c * total + e] + mean[c], 255.f);
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);  // clamp_(0, 1); NaN stays NaN -> byte 0 below
            const float s = v * 255.f;
            out[e * 3 + (2 - c)] = s != s ? (unsigned char)0 : (unsigned char)s;
        }
    }
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_channel_stats_workspace_bytes(int h, int w) {
    if (!conv_dims_ok(1, 3, h, w, 3, 0)) return 0;
    const size_t tiles = (size_t)((h + ST_ROWS - 1) / ST_ROWS) * ((w + ST_COLS - 1) / ST_COLS);
    return tiles * 9 * sizeof(double);
}

int maua_channel_stats(const float* x_bchw, const float* noise_bwhc, float noise_amp, int frames, int slot, int h, int w,
                       double* stats9, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x_bchw && stats9 && workspace, MAUA_E_INVAL, "channel_stats: null pointer");
    MAUA_REQUIRE(conv_dims_ok(1, 3, h, w, 3, 0), MAUA_E_INVAL, "channel_stats: bad dims %d x %d", h, w);
    MAUA_REQUIRE(frames > 0 && slot >= 0 && slot < frames, MAUA_E_INVAL, "channel_stats: slot %d of %d frames", slot, frames);
    MAUA_REQUIRE(workspace_bytes >= maua_channel_stats_workspace_bytes(h, w), MAUA_E_WORKSPACE, "channel_stats: workspace too small");
    const int tiles_w = (w + ST_COLS - 1) / ST_COLS, tiles_h = (h + ST_ROWS - 1) / ST_ROWS;
    hipLaunchKernelGGL(channel_stats_kernel, dim3(tiles_w * tiles_h), dim3(256), 0, (hipStream_t)stream, x_bchw, noise_bwhc, noise_amp, frames,
                       slot, h, w, tiles_w, (double*)workspace);
    int rc = check_launch("channel_stats_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(channel_stats_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace,
                       tiles_w * tiles_h, stats9);
    return check_launch("channel_stats_finish_kernel");
}

int maua_color_match_solve(const double* stats_target, int frames, int64_t pixels_target, const double* stats_source,
                           int64_t pixels_source, float eps, float* coef16, maua_stream_t stream) {
    MAUA_REQUIRE(stats_target && stats_source && coef16, MAUA_E_INVAL, "color_match_solve: null pointer");
    MAUA_REQUIRE(frames > 0 && pixels_target > 0 && pixels_source > 0, MAUA_E_INVAL, "color_match_solve: empty image");
    hipLaunchKernelGGL(color_match_solve_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats_target, frames, (double)pixels_target,
                       stats_source, (double)pixels_source, (double)eps, coef16);
    return check_launch("color_match_solve_kernel");
}

int maua_color_match_apply(const float* x_bchw, const float* noise_bwhc, float noise_amp, const float* coef16, const float* all_coef,
                           int n_coef, float weight, int accumulate, int frames, int slot, int h, int w, float* out_bchw,
                           maua_stream_t stream) {
    MAUA_REQUIRE(x_bchw && coef16 && all_coef && out_bchw, MAUA_E_INVAL, "color_match_apply: null pointer");
    MAUA_REQUIRE(conv_dims_ok(1, 3, h, w, 3, 0) && n_coef > 0, MAUA_E_INVAL, "color_match_apply: bad dims");
    MAUA_REQUIRE(frames > 0 && slot >= 0 && slot < frames, MAUA_E_INVAL, "color_match_apply: slot %d of %d frames", slot, frames);
    const int tiles_w = (w + ST_COLS - 1) / ST_COLS, tiles_h = (h + ST_ROWS - 1) / ST_ROWS;
    hipLaunchKernelGGL(color_match_apply_kernel, dim3(tiles_w * tiles_h), dim3(256), 0, (hipStream_t)stream, x_bchw, noise_bwhc, noise_amp,
                       coef16, all_coef, n_coef, weight, accumulate, frames, slot, h, w, tiles_w, out_bchw);
    return check_launch("color_match_apply_kernel");
}

int maua_resize_bilinear(const float* x, float* y, int planes, int h, int w, int oh, int ow, float scale_h, float scale_w,
                         maua_stream_t stream) {
    MAUA_REQUIRE(x && y, MAUA_E_INVAL, "resize_bilinear: null pointer");
    MAUA_REQUIRE(conv_dims_ok(1, planes, h, w, 1, 0) && conv_dims_ok(1, planes, oh, ow, 1, 0) && scale_h > 0.f && scale_w > 0.f, MAUA_E_INVAL,
                 "resize_bilinear: bad dims");
    const int64_t total = (int64_t)planes * oh * ow;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(reduce_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, planes, h, w, oh,
                       ow, scale_h, scale_w);
    return check_launch("resize_bilinear_kernel");
}

int maua_deprocess_u8(const float* x_bgr_chw, unsigned char* out_rgb_hwc, int h, int w, float mean_b, float mean_g, float mean_r,
                      maua_stream_t stream) {
    MAUA_REQUIRE(x_bgr_chw && out_rgb_hwc, MAUA_E_INVAL, "deprocess_u8: null pointer");
    MAUA_REQUIRE(conv_dims_ok(1, 3, h, w, 3, 0), MAUA_E_INVAL, "deprocess_u8: bad dims");
    hipLaunchKernelGGL(deprocess_u8_kernel, dim3(reduce_blocks((int64_t)h * w, 256)), dim3(256), 0, (hipStream_t)stream, x_bgr_chw,
                       out_rgb_hwc, h, w, mean_b, mean_g, mean_r);
    return check_launch("deprocess_u8_kernel");
}

}  // extern "C"
