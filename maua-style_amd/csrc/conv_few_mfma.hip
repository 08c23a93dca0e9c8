// The backward-data pass of the image layer on the matrix cores (round 5): 64 gradient channels -> 3 pixel channels,
// autograd's conv_backward of `nn.Conv2d(3, 64, 3, padding=1)` (/root/reference/models.py:129-130).
//
//     gx[c][i][j] = sum_k sum_(ky,kx) gy[k][i - ky + 1][j - kx + 1] w[k][c][ky][kx]
//
// As a matrix product with pixels x 3 outputs the K = 576 sum fills 3 of 16 MFMA columns.  The useful orientation is per INPUT pixel p:
//
//     T[p][(ky, kx, c)] = sum_k gy[k][p] w[k][c][ky][kx]          M = 27 (tap, channel) columns of 32, N = 32 pixels, K = 64
//     gx[c][q]          = sum_(ky,kx) T[q - (ky - 1, kx - 1)][(ky, kx, c)]     nine values gathered from the neighbours
//
// 24 `v_mfma_f32_32x32x16_bf16` per 32 pixels in the exact bf16x6 arithmetic of conv_img.hip / conv_x6.hip (three bf16 parts per operand,
// the six products that reach 2^-24: no per-chunk scale to compute) = 26 GFLOP of 16-bit matrix work at 1024 x 1024 beside the 268 MB that
// have to be read; conv3x3_few_out_kernel (conv_direct.hip) issues 1.8 G packed fp32 FMAs for the same sums.
//
// Workgroup = 4 waves = ROWS x 62 output pixels: the T values of the (ROWS + 2) x 64 gradient pixels around them (two 32-pixel MFMA blocks
// per row, the blocks dealt over the waves, the next block's 32 loads in flight under the current block's products) go to LDS as
// T[27][(ROWS + 2) x 64] floats, then every output pixel adds its nine values per channel in tap order.  Gradient pixels outside the image
// are read as 0 through the buffer descriptor's range check.  Filters: 12 fragments of 16 bytes per lane (bank packed once per weight set),
// in registers for the wave's life.
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace maua {

typedef __attribute__((ext_vector_type(8))) __bf16 fm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fm_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int fm_u32x4;
typedef float fm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fm_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short fm_bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float fm_bf16_value(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned fm_cvt_pk(float a, float b) {
    const fm_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, fm_bf16x2));
}

constexpr int FM_K = 64;                  // consumed channels (the image layer's 64 filters)
constexpr int FM_STEPS = FM_K / 16;
constexpr int FM_TC = 64, FM_OC = 62;     // T columns / output columns of a tile
constexpr int FM_BANK_BYTES = FM_STEPS * 3 * 64 * 16;

// bank[step][part][lane][8]: lane = (m = lane % 32, K group = lane / 32) holds part `part` of A[m][k = 16 step + 8 group + i],
// A[(ky 3 + kx) co + c][k] = w[k][c][ky][kx] (OIHW, un-flipped: the gather applies the shift), zero for m >= 9 co.
__global__ void pack_few_mfma_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int co) {
    const int total = FM_STEPS * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e % 8, lane = (e / 8) % 64, step = e / 512;
        const int m = lane & 31, k = 16 * step + 8 * (lane >> 5) + i;
        float v = 0.f;
        if (m < 9 * co) {
            const int tap = m / co, c = m - tap * co;
            v = w[((int64_t)k * co + c) * 9 + tap];
        }
        const unsigned short h = fm_bf16_bits(v);
        const float r1 = v - fm_bf16_value(h);
        const unsigned short md = fm_bf16_bits(r1);
        const unsigned short l = fm_bf16_bits(r1 - fm_bf16_value(md));
        const unsigned short parts[3] = {h, md, l};
        for (int part = 0; part < 3; ++part) bank[(((int64_t)step * 3 + part) * 64 + lane) * 8 + i] = parts[part];
    }
}

struct FewMfmaArgs {
    const float* gy;
    const unsigned char* bank;
    float* gx;
    int H, W, co;
    int tiles_x, tiles, per_xcd;  // grid x = 8 per_xcd workgroups: workgroup id takes tile (id % 8) per_xcd + id / 8 - XCD k (ids = k mod 8) walks
                                  // the k-th contiguous band of the row-major tile list, its CUs side by side (conv_x3w.hip's order; the
                                  // dispatch order - tile = id - reads the same bytes at half the rate: tools/mfma_probe/stage_bw.hip)
};

template <int ROWS>
__global__ void __launch_bounds__(256, 2) conv_few_mfma_kernel(FewMfmaArgs p) {
    extern __shared__ float T[];  // [27][TR x 64]
    constexpr int TR = ROWS + 2, TPL = TR * FM_TC, NBLK = TR * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, kg = lane >> 5;
    const int tile = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= p.tiles) return;  // (whole workgroup)
    const int ty = tile / p.tiles_x;
    const int x0 = (tile - ty * p.tiles_x) * FM_OC, y0 = ty * ROWS, img = blockIdx.z;
    const int64_t plane = (int64_t)p.H * p.W;
    fm_bf16x8 a[FM_STEPS][3];
#pragma unroll
    for (int step = 0; step < FM_STEPS; ++step)
#pragma unroll
        for (int part = 0; part < 3; ++part)
            a[step][part] = *reinterpret_cast<const fm_bf16x8*>(p.bank + (((step * 3 + part) * 64) + lane) * 16);
    // the image's 64 gradient planes behind one descriptor: a pixel outside the plane gets an offset beyond its range and reads 0
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy) + (int64_t)img * FM_K * plane, 0,
                                                                        (unsigned)(FM_K * plane * 4), 0x00020000);
    auto request = [&](float (&v)[FM_STEPS][8], int b) {
        const int tr = b >> 1, Y = y0 - 1 + tr, X = x0 - 1 + 32 * (b & 1) + nl;
        const bool ok = Y >= 0 && Y < p.H && X >= 0 && X < p.W;
        // (the lane half's eight channels ride in the vector offset, the step's and the register's channel in the scalar offset, which
        // the range check does not see: an out-of-image pixel is out of range whatever the channel)
        const unsigned voff = ok ? (unsigned)(((int64_t)kg * 8 * plane + (int64_t)Y * p.W + X) * 4) : 0x80000000u;
#pragma unroll
        for (int step = 0; step < FM_STEPS; ++step)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[step][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)((16 * step + i) * plane * 4), 0));
    };
    float v[FM_STEPS][8], vn[FM_STEPS][8];
    if (wave < NBLK) request(v, wave);
    for (int b = wave; b < NBLK; b += 4) {
        if (b + 4 < NBLK) request(vn, b + 4);
        fm_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int step = 0; step < FM_STEPS; ++step) {
            fm_u32x4 bp[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float u0 = v[step][2 * q], u1 = v[step][2 * q + 1];
                const unsigned h = fm_cvt_pk(u0, u1);
                const float r0 = u0 - __builtin_bit_cast(float, h << 16), r1 = u1 - __builtin_bit_cast(float, h & 0xffff0000u);
                const unsigned m = fm_cvt_pk(r0, r1);
                const unsigned l = fm_cvt_pk(r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u));
                bp[0][q] = h;
                bp[1][q] = m;
                bp[2][q] = l;
            }
            const fm_bf16x8 b0v = __builtin_bit_cast(fm_bf16x8, bp[0]), b1v = __builtin_bit_cast(fm_bf16x8, bp[1]),
                            b2v = __builtin_bit_cast(fm_bf16x8, bp[2]);
            // smallest products first (conv_x6.hip's order)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][2], b0v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][1], b1v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][0], b2v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][1], b0v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][0], b1v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step][0], b0v, acc, 0, 0, 0);
        }
        // register r of lane (pixel nl, half kg) = column m = (r & 3) + 8 (r >> 2) + 4 kg
        float* dst = T + (b >> 1) * FM_TC + 32 * (b & 1) + nl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (m < 9 * p.co) dst[m * TPL] = acc[r];
        }
#pragma unroll
        for (int step = 0; step < FM_STEPS; ++step)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[step][i] = vn[step][i];
    }
    __syncthreads();
    const int co = p.co;
    for (int idx = tid; idx < ROWS * FM_OC; idx += 256) {
        const int i = idx / FM_OC, j = idx - i * FM_OC;
        const int Y = y0 + i, X = x0 + j;
        if (Y >= p.H || X >= p.W) continue;
        const float* t0 = T + (i + 2) * FM_TC + j + 2;
        for (int c = 0; c < co; ++c) {
            float s = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) s += t0[((ky * 3 + kx) * co + c) * TPL - ky * FM_TC - kx];
            p.gx[((int64_t)img * co + c) * plane + (int64_t)Y * p.W + X] = s;
        }
    }
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_few_mfma_bank_bytes(void) { return FM_BANK_BYTES; }

int maua_conv_pack_filters_few_mfma(const float* w_oihw, void* bank, int cout, int cin, maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && bank, MAUA_E_INVAL, "conv_pack_filters_few_mfma: null pointer");
    MAUA_REQUIRE(cout == FM_K && cin >= 1 && cin <= 3, MAUA_E_UNSUPPORTED, "conv_pack_filters_few_mfma: a 3x3 layer with %d filters over 1-3 channels (got %d over %d)", FM_K, cout, cin);
    hipLaunchKernelGGL(pack_few_mfma_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank, cin);
    return check_launch("pack_few_mfma_kernel");
}

// Whether maua_conv3x3_few_mfma takes the backward-data pass of a 3x3, stride-1, padding-`pad` layer with `cout` filters over `cin` channels on
// an h x w image (the gradient has the image's size: padding 1)
int maua_conv_few_mfma_supported(int n, int cin, int h, int w, int cout, int pad) {
    return conv_dims_ok(n, cin, h, w, cout, pad) && cout == FM_K && cin >= 1 && cin <= 3 && pad == 1 && n <= 65535 && (int64_t)FM_K * h * w * 4 < (1ll << 31) &&
           (h + 3) / 4 <= 65535;
}

// tile: 0 = the library's choice (4 output rows per workgroup below a million pixels, 8 from there); 1 / 2 / 3 = 4 / 8 / 14 rows x 62 columns
int maua_conv3x3_few_mfma(const float* gy, const void* bank, float* gx, int n, int cin, int h, int w, int cout, int tile, maua_stream_t stream) {
    MAUA_REQUIRE(gy && bank && gx, MAUA_E_INVAL, "conv3x3_few_mfma: null pointer");
    MAUA_REQUIRE(maua_conv_few_mfma_supported(n, cin, h, w, cout, 1), MAUA_E_UNSUPPORTED, "conv3x3_few_mfma: unsupported geometry");
    MAUA_REQUIRE(tile >= 0 && tile <= 3, MAUA_E_INVAL, "conv3x3_few_mfma: tile code 0 - 3");
    if (tile == 0) tile = (int64_t)h * w < 1000000 ? 1 : 2;
    FewMfmaArgs p{gy, (const unsigned char*)bank, gx, h, w, cin, 0, 0, 0};
    static unsigned long long attr[4] = {0, 0, 0, 0};
    const int rows = tile == 1 ? 4 : tile == 2 ? 8 : 14;
    const int lds = 27 * (rows + 2) * FM_TC * 4;
    p.tiles_x = (w + FM_OC - 1) / FM_OC;
    p.tiles = p.tiles_x * ((h + rows - 1) / rows);
    p.per_xcd = (p.tiles + 7) / 8;
    const dim3 grid((unsigned)(8 * p.per_xcd), 1, (unsigned)n);
    if (rows == 4) {
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(conv_few_mfma_kernel<4>), lds, &attr[1]);
        hipLaunchKernelGGL(conv_few_mfma_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else if (rows == 8) {
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(conv_few_mfma_kernel<8>), lds, &attr[2]);
        hipLaunchKernelGGL(conv_few_mfma_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(conv_few_mfma_kernel<14>), lds, &attr[3]);
        hipLaunchKernelGGL(conv_few_mfma_kernel<14>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    return check_launch("conv_few_mfma_kernel");
}

}  // extern "C"
