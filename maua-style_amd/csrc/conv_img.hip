// The image layer: a 3x3 stride-1 convolution that consumes at most three channels (conv1_1 of every VGG, `nn.Conv2d(3, 64, 3, padding=1)`
// + `nn.ReLU`, /root/reference/models.py:129-130) in the exact bf16x6 arithmetic of conv_x6.hip - three bf16 parts per operand, the six
// products that reach 2^-24, fp32 accumulation on the matrix cores.
//
// conv_x6.hip treats the image as one 8-channel chunk of a general layer: K = 72 per output, 45 matrix instructions per 32 x 32 block for 27
// products, patch and filters through LDS.  Here K is what it is - the 27 (channel, tap) pairs, padded to two K = 16 steps - so a 64-channel
// x 32-pixel block is 24 matrix instructions (768 cycles) for 8 KB of output: the kernel is bound by WRITING the activation (268 MB at
// 1024 x 1024), not by arithmetic.  No LDS at all:
//   A (filters): a lane's fragments - [32-channel block][K step][part] x 8 bf16 - are 12 x 16 bytes from a bank packed once per weight set
//                and stay in registers for the wave's whole life;
//   B (pixels):  lane (pixel n = lane % 32, K group = lane / 32) gathers its sixteen (channel, tap) values of the block straight from the
//                image (12 MB at 1024 x 1024: L2-resident, neighbouring lanes read neighbouring pixels) and splits them itself;
//   bias: the first free pair of the padded K carries it (filter value = bias, pixel value = 1): added exactly by the matrix unit;
//   a wave walks blocks of 32 pixels of one row in a grid-stride loop; ReLU in registers, 128-byte row segments out.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace maua {

typedef __attribute__((ext_vector_type(8))) __bf16 ci_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ci_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int ci_u32x4;
typedef float ci_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ci_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short ci_bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }  // round to nearest even
__device__ __forceinline__ float ci_bf16_value(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned ci_cvt_pk(float a, float b) {
    const ci_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ci_bf16x2));
}

constexpr int CI_KPAD = 32;                                   // 27 (channel, tap) pairs in two K = 16 steps
#ifndef CI_OCC
#define CI_OCC 2                                               // workgroups per CU the register budget is set for (3: spills, slower on small images; 4: half the speed)
#endif
constexpr int CI_TILE_BYTES = 2 * 2 * 3 * 64 * 16;            // one 64-channel tile of the bank: [block32][step][part][lane][16 B] = 12 KiB

// bank[tile][block32][step][part][lane][8]: lane = (m = lane % 32, K group = lane / 32) holds part `part` of w[co = tile 64 + block 32 + m]
// [k = 16 step + 8 group + i], k = ci 9 + ky 3 + kx; k = 9 cin carries the BIAS (the kernel feeds that pair the value 1: the three parts
// of the bias meet bf16(1) in three of the six products and add up to the bias exactly); zero beyond and for co >= cout.
__global__ void pack_image_kernel(const float* __restrict__ w, const float* __restrict__ bias, unsigned short* __restrict__ bank, int cout,
                                  int cin) {
    const int ntile = (cout + 63) / 64;
    const int64_t total = (int64_t)ntile * 2 * 2 * 64 * 8;  // (tile, block, step, lane, i): all three parts at once
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int i = (int)(r % 8);
        r /= 8;
        const int lane = (int)(r % 64);
        r /= 64;
        const int step = (int)(r % 2);
        r /= 2;
        const int blk = (int)(r % 2);
        const int tile = (int)(r / 2);
        const int co = tile * 64 + blk * 32 + (lane & 31), k = 16 * step + 8 * (lane >> 5) + i;
        float v = 0.f;
        if (co < cout && k < 9 * cin) v = w[(int64_t)co * cin * 9 + k];  // OIHW: [co][ci][ky][kx] = [co][k]
        if (co < cout && k == 9 * cin && bias) v = bias[co];
        const unsigned short h = ci_bf16_bits(v);
        const float r1 = v - ci_bf16_value(h);
        const unsigned short m = ci_bf16_bits(r1);
        const unsigned short l = ci_bf16_bits(r1 - ci_bf16_value(m));
        const unsigned short parts[3] = {h, m, l};
        for (int part = 0; part < 3; ++part)
            bank[(((((int64_t)tile * 2 + blk) * 2 + step) * 3 + part) * 64 + lane) * 8 + i] = parts[part];
    }
}

struct ImgArgs {
    const float* x;
    const unsigned char* bank;
    float* y;
    int n, cin, H, W, cout, OH, OW, pad, relu;
    int blocks_x;      // 32-pixel blocks per output row
    int64_t blocks;    // n * OH * blocks_x
};

__global__ void __launch_bounds__(256, CI_OCC) conv_image_kernel(ImgArgs p) {
    const int lane = threadIdx.x & 63;
    const int nl = lane & 31, kg = lane >> 5;
    const int tile = blockIdx.y;
    // filters of this lane: [block32][step][part]
    ci_bf16x8 a[2][2][3];
    {
        const unsigned char* src = p.bank + (int64_t)tile * CI_TILE_BYTES + lane * 16;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int step = 0; step < 2; ++step)
#pragma unroll
                for (int part = 0; part < 3; ++part)
                    a[blk][step][part] = *reinterpret_cast<const ci_bf16x8*>(src + (((blk * 2 + step) * 3 + part) * 64) * 16);
    }
    // the sixteen (channel, tap) pairs of this lane: offset from the pixel's own element of plane 0, and the tap's row / column shift
    const int64_t plane = (int64_t)p.H * p.W;
    const int kmax = 9 * p.cin;
    int rel[2][8];
    unsigned ones = 0;  // bit 8 step + i: that pair is the bias pair (value 1)
#pragma unroll
    for (int step = 0; step < 2; ++step)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int k = 16 * step + 8 * kg + i;
            if (k == kmax) ones |= 1u << (8 * step + i);
            k = k < kmax ? k : 0;  // (padding pairs meet zero filter values: any finite image value will do)
            const int ci = k / 9, ky = (k - 9 * ci) / 3, kx = k - 9 * ci - 3 * ky;
            rel[step][i] = (int)(ci * plane) + (ky - p.pad) * p.W + (kx - p.pad);
        }
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    // a wave's blocks are CONTIGUOUS in (image, row, 32-pixel block) order: one decomposition at the start, then increments (no division per
    // block), and its gathers walk along rows.  The values of block t + 1 are requested before block t is multiplied.
    const int nw = (int)gridDim.x * 4;
    const int wid = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    const int per = (int)((p.blocks + nw - 1) / nw);
    const int b0 = wid * per, b1 = (int)(p.blocks < (int64_t)b0 + per ? p.blocks : (int64_t)b0 + per);
    if (b0 >= b1) return;
    int bx = b0 % p.blocks_x, oy = (b0 / p.blocks_x) % p.OH, img = (b0 / p.blocks_x) / p.OH;
    auto gather = [&](float (&v)[2][8], int gi, int gy, int gb) {
        const int ox = gb * 32 + nl;
        const float* __restrict__ xin = p.x + (int64_t)gi * p.cin * plane;
        const int centre = gy * p.W + ox;  // input element under the output pixel at shift (0, 0)  (cin H W < 2^31)
        const bool interior = gy - p.pad >= 0 && gy - p.pad + 2 < p.H && gb * 32 - p.pad >= 0 && gb * 32 + 31 - p.pad + 2 < p.W;  // wave-uniform
        if (interior) {
#pragma unroll
            for (int step = 0; step < 2; ++step)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[step][i] = xin[centre + rel[step][i]];
        } else {
#pragma unroll
            for (int step = 0; step < 2; ++step)
#pragma unroll
                for (int i = 0; i < 8; ++i) {  // (image borders only: the tap's shift recomputed rather than kept in registers)
                    int k = 16 * step + 8 * kg + i;
                    k = k < kmax ? k : 0;
                    const int t9 = k % 9;
                    const int iy = gy + t9 / 3 - p.pad, ix = ox + t9 % 3 - p.pad;
                    const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    v[step][i] = ok ? xin[centre + rel[step][i]] : 0.f;
                }
        }
    };
    float v[2][8], vn[2][8];
    gather(v, img, oy, bx);
    for (int b = b0; b < b1; ++b) {
        bx = __builtin_amdgcn_readfirstlane(bx);  // (wave-uniform by construction; said so, the descriptor below stays scalar)
        oy = __builtin_amdgcn_readfirstlane(oy);
        img = __builtin_amdgcn_readfirstlane(img);
        int nbx = bx + 1, noy = oy, nimg = img;
        if (nbx == p.blocks_x) {
            nbx = 0;
            if (++noy == p.OH) {
                noy = 0;
                ++nimg;
            }
        }
        if (b + 1 < b1) gather(vn, nimg, noy, nbx);
#pragma unroll
        for (int step = 0; step < 2; ++step)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[step][i] = (ones >> (8 * step + i)) & 1u ? 1.f : v[step][i];
        // three bf16 parts of every value, pairs packed
        ci_u32x4 bp[2][3];
#pragma unroll
        for (int step = 0; step < 2; ++step)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = v[step][2 * q], x1 = v[step][2 * q + 1];
                const unsigned h = ci_cvt_pk(x0, x1);
                const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
                const unsigned m = ci_cvt_pk(r0, r1);
                const unsigned l = ci_cvt_pk(r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u));
                bp[step][0][q] = h;
                bp[step][1][q] = m;
                bp[step][2][q] = l;
            }
        ci_f32x16 acc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
#pragma unroll
        for (int step = 0; step < 2; ++step) {  // smallest products first (conv_x6.hip's order); the two channel blocks alternate: no
                                                // instruction waits for the one before it
            const ci_bf16x8 b0v = __builtin_bit_cast(ci_bf16x8, bp[step][0]), b1v = __builtin_bit_cast(ci_bf16x8, bp[step][1]),
                            b2v = __builtin_bit_cast(ci_bf16x8, bp[step][2]);
#define CI_PAIR(AP, BV)                                                                                         \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][step][AP], BV, acc[0], 0, 0, 0);                       \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][step][AP], BV, acc[1], 0, 0, 0);
            CI_PAIR(2, b0v) CI_PAIR(1, b1v) CI_PAIR(0, b2v) CI_PAIR(1, b0v) CI_PAIR(0, b1v) CI_PAIR(0, b0v)
#undef CI_PAIR
        }
        // stores: the lane's pixel as vector offset, a register's channel as scalar offset (computed, not held: 32 of them), lanes past the
        // row's end and channels past cout out of the descriptor's range
        const int ox = bx * 32 + nl;
        const int tile_ch = min(64, p.cout - tile * 64);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.y + ((int64_t)img * p.cout + tile * 64) * out_plane, 0,
                                                                            (unsigned)(tile_ch * out_plane * 4), 0x00020000);
        const unsigned voff = ox < p.OW ? (unsigned)((4 * kg * out_plane + (int64_t)oy * p.OW + ox) * 4) : 0x80000000u;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = blk * 32 + (r & 3) + 8 * (r >> 2);
                float o = acc[blk][r];
                if (p.relu) o = o > 0.f ? o : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, voff, (int)(cr * out_plane * 4), 0);
            }
#pragma unroll
        for (int step = 0; step < 2; ++step)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[step][i] = vn[step][i];
        bx = nbx;
        oy = noy;
        img = nimg;
    }
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_image_bank_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin > 3 || cout > (1 << 16)) return 0;
    return (size_t)((cout + 63) / 64) * CI_TILE_BYTES;
}

int maua_conv_pack_filters_image(const float* w_oihw, const float* bias, void* bank, int cout, int cin, maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && bank && cout > 0 && cout <= (1 << 16) && cin > 0 && cin <= 3, MAUA_E_INVAL,
                 "conv_pack_filters_image: needs 1-3 input channels");
    hipLaunchKernelGGL(pack_image_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w_oihw, bias, (unsigned short*)bank, cout, cin);
    return check_launch("pack_image_kernel");
}

int maua_conv3x3_image(const float* x, const void* bank, float* y, int n, int cin, int h, int w, int cout, int pad, int relu,
                       maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y, MAUA_E_INVAL, "conv3x3_image: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2 && cin <= 3, MAUA_E_INVAL, "conv3x3_image: bad dims (1-3 input channels)");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_image: input smaller than the filter");
    MAUA_REQUIRE((int64_t)cin * h * w < (1ll << 31), MAUA_E_UNSUPPORTED, "conv3x3_image: image too large");
    ImgArgs p{};
    p.x = x;
    p.bank = (const unsigned char*)bank;
    p.y = y;
    p.n = n;
    p.cin = cin;
    p.H = h;
    p.W = w;
    p.cout = cout;
    p.OH = h + 2 * pad - 2;
    p.OW = w + 2 * pad - 2;
    p.pad = pad;
    p.relu = relu;
    p.blocks_x = (p.OW + 31) / 32;
    p.blocks = (int64_t)n * p.OH * p.blocks_x;
    MAUA_REQUIRE(p.blocks < (1ll << 31) && (int64_t)64 * p.OH * p.OW < (1ll << 30), MAUA_E_UNSUPPORTED, "conv3x3_image: plane too large");
    const int64_t want = (p.blocks + 3) / 4;  // workgroups of four waves, a block per wave at least; at most three per CU (what the registers allow)
    const unsigned gx = (unsigned)(want < 256 * CI_OCC ? (want > 0 ? want : 1) : 256 * CI_OCC);
    hipLaunchKernelGGL(conv_image_kernel, dim3(gx, (unsigned)((cout + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("conv_image_kernel");
}

}  // extern "C"
