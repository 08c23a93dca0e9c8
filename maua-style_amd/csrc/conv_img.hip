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
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
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
    float* gram_slabs;  // GRAM: gridDim.x slabs of 64 x 64 floats
    int n, cin, H, W, cout, OH, OW, pad, relu;
    int blocks_x;      // 32-pixel blocks per output row
    int64_t blocks;    // n * OH * blocks_x
};

// GRAM (single image, one 64-channel tile, ReLU): every workgroup also leaves the 64 x 64 partial sum of Y Y^T over ITS pixels (Y = the
// activation it writes) as one split-K slab of gram.hip's finishing kernels - the Gram matrix of relu1_1 without reading the 268 MB
// back.  The product needs the tile with channels on the lanes and pixels in the registers: exchanging the two operands of the same
// matrix instructions yields exactly that (24 more instructions on a kernel bound by its stores), and an accumulator tile whose
// ROW index is summed over is its own operand (cdna_hip_programming.md, 'an accumulator tile as the next MFMA's operand'): no LDS, no
// lane movement.  bf16 triples again (no scales), fp32 sums over the wave's ~10 blocks, the four waves added in wave order.
template <bool GRAM>
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
    ci_f32x16 gsum[3];  // GRAM: channel blocks (0,0), (0,1), (1,1) of the partial Gram matrix
    if constexpr (GRAM) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) gsum[q][r] = 0.f;
    }
    if (!GRAM && b0 >= b1) return;
    int bx = b0 < b1 ? b0 % p.blocks_x : 0, oy = b0 < b1 ? (b0 / p.blocks_x) % p.OH : 0, img = b0 < b1 ? (b0 / p.blocks_x) / p.OH : 0;
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
    if (b0 < b1) gather(v, img, oy, bx);
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
        if constexpr (GRAM) {
            ci_f32x16 xt[2];  // the tile transposed: lane = channel (of the block), register = pixel
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) xt[blk][r] = 0.f;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                const ci_bf16x8 b0v = __builtin_bit_cast(ci_bf16x8, bp[step][0]), b1v = __builtin_bit_cast(ci_bf16x8, bp[step][1]),
                                b2v = __builtin_bit_cast(ci_bf16x8, bp[step][2]);
#define CI_PAIR_T(AP, BV)                                                                                       \
    xt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BV, a[0][step][AP], xt[0], 0, 0, 0);                         \
    xt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BV, a[1][step][AP], xt[1], 0, 0, 0);
                CI_PAIR_T(2, b0v) CI_PAIR_T(1, b1v) CI_PAIR_T(0, b2v) CI_PAIR_T(1, b0v) CI_PAIR_T(0, b1v) CI_PAIR_T(0, b0v)
#undef CI_PAIR_T
            }
            // ReLU, pixels past the end of the row out, bf16 triples: registers 8 s ... 8 s + 7 are the fragment of K step s
            ci_u32x4 fr[2][2][3];  // [channel block][K step][part]
            const int px0 = bx * 32 + 4 * kg;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int sK = 0; sK < 2; ++sK)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r0i = 8 * sK + 2 * q, r1i = r0i + 1;
                        float x0 = xt[blk][r0i], x1 = xt[blk][r1i];
                        x0 = (x0 > 0.f && px0 + (r0i & 3) + 8 * (r0i >> 2) < p.OW) ? x0 : 0.f;
                        x1 = (x1 > 0.f && px0 + (r1i & 3) + 8 * (r1i >> 2) < p.OW) ? x1 : 0.f;
                        const unsigned h = ci_cvt_pk(x0, x1);
                        const float e0 = x0 - __builtin_bit_cast(float, h << 16), e1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
                        const unsigned m = ci_cvt_pk(e0, e1);
                        const unsigned l = ci_cvt_pk(e0 - __builtin_bit_cast(float, m << 16), e1 - __builtin_bit_cast(float, m & 0xffff0000u));
                        fr[blk][sK][0][q] = h;
                        fr[blk][sK][1][q] = m;
                        fr[blk][sK][2][q] = l;
                    }
#pragma unroll
            for (int sK = 0; sK < 2; ++sK) {
#define CI_G(PA, PB)                                                                                                                        \
    gsum[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ci_bf16x8, fr[0][sK][PA]), __builtin_bit_cast(ci_bf16x8, fr[0][sK][PB]), gsum[0], 0, 0, 0); \
    gsum[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ci_bf16x8, fr[0][sK][PA]), __builtin_bit_cast(ci_bf16x8, fr[1][sK][PB]), gsum[1], 0, 0, 0); \
    gsum[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ci_bf16x8, fr[1][sK][PA]), __builtin_bit_cast(ci_bf16x8, fr[1][sK][PB]), gsum[2], 0, 0, 0);
                CI_G(2, 0) CI_G(1, 1) CI_G(0, 2) CI_G(1, 0) CI_G(0, 1) CI_G(0, 0)
#undef CI_G
            }
        }
#pragma unroll
        for (int step = 0; step < 2; ++step)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[step][i] = vn[step][i];
        bx = nbx;
        oy = noy;
        img = nimg;
    }
    if constexpr (GRAM) {
        // waves 1-3 leave their sums in LDS, wave 0 adds them in wave order and writes the workgroup's slab: element (row co_i, column co_j)
        // of the 64 x 64 tile, upper channel blocks only (gram.hip's finishing kernels read the lower triangle from the upper one)
        __shared__ float red[3][3][16][64];
        const int wave = threadIdx.x >> 6;
        if (wave > 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][q][r][lane] = gsum[q][r];
        }
        __syncthreads();
        if (wave > 0) return;
        float* __restrict__ slab = p.gram_slabs + (int64_t)blockIdx.x * 64 * 64;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int bi = q == 2 ? 1 : 0, bj = q == 0 ? 0 : 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float g = gsum[q][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) g += red[w][q][r][lane];
                slab[(bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg) * 64 + bj * 32 + nl] = g;
            }
        }
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

// What the kernel's 32-bit offsets hold: an image below 2^31 elements, fewer than 2^31 row blocks, and 64 output planes below 2^32 bytes (the
// store descriptor spans a 64-channel tile: planes of fewer than 2^24 pixels - a 4096 x 4096 image is one pixel too many).
static bool conv_image_fits(int64_t n, int64_t cin, int64_t h, int64_t w, int64_t pad) {
    const int64_t oh = h + 2 * pad - 2, ow = w + 2 * pad - 2;
    return cin * h * w < (1ll << 31) && n * oh * ((ow + 31) / 32) < (1ll << 31) && 64 * oh * ow < (1ll << 30);
}

static int conv_image_fill(ImgArgs& p, const float* x, const void* bank, float* y, int n, int cin, int h, int w, int cout, int pad, int relu) {
    MAUA_REQUIRE(x && bank && y, MAUA_E_INVAL, "conv3x3_image: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2 && cin <= 3, MAUA_E_INVAL, "conv3x3_image: bad dims (1-3 input channels)");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_image: input smaller than the filter");
    MAUA_REQUIRE(conv_image_fits(n, cin, h, w, pad), MAUA_E_UNSUPPORTED, "conv3x3_image: plane too large (maua_conv_image_supported)");
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
    return MAUA_OK;
}
// workgroups of four waves, a block per wave at least; at most CI_OCC per CU (what the registers allow)
static unsigned conv_image_grid(int64_t blocks) {
    const int64_t want = (blocks + 3) / 4;
    return (unsigned)(want < 256 * CI_OCC ? (want > 0 ? want : 1) : 256 * CI_OCC);
}

int maua_conv3x3_image(const float* x, const void* bank, float* y, int n, int cin, int h, int w, int cout, int pad, int relu,
                       maua_stream_t stream) {
    ImgArgs p{};
    const int rc = conv_image_fill(p, x, bank, y, n, cin, h, w, cout, pad, relu);
    if (rc) return rc;
    hipLaunchKernelGGL(conv_image_kernel<false>, dim3(conv_image_grid(p.blocks), (unsigned)((cout + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("conv_image_kernel");
}

int maua_conv_image_supported(int n, int cin, int h, int w, int cout, int pad) {
    return conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2 && cin <= 3 && h + 2 * pad >= 3 && w + 2 * pad >= 3 && conv_image_fits(n, cin, h, w, pad);
}

int maua_conv_image_gram_slabs(int h, int w, int pad) {
    if (h <= 0 || w <= 0 || pad < 0 || pad > 2 || h + 2 * pad < 3 || w + 2 * pad < 3 || !conv_image_fits(1, 3, h, w, pad)) return 0;
    const int64_t oh = h + 2 * pad - 2, ow = w + 2 * pad - 2;
    return (int)conv_image_grid(oh * ((ow + 31) / 32));
}

int maua_conv3x3_image_gram(const float* x, const void* bank, float* y, float* gram_slabs, int cin, int h, int w, int pad, maua_stream_t stream) {
    MAUA_REQUIRE(gram_slabs, MAUA_E_INVAL, "conv3x3_image_gram: null slabs");
    ImgArgs p{};
    const int rc = conv_image_fill(p, x, bank, y, 1, cin, h, w, 64, pad, 1);
    if (rc) return rc;
    p.gram_slabs = gram_slabs;
    hipLaunchKernelGGL(conv_image_kernel<true>, dim3(conv_image_grid(p.blocks), 1), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("conv_image_kernel");
}

}  // extern "C"
