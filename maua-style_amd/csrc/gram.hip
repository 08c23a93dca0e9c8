// Gram / covariance matrix G = scale * Fc Fc^T of a feature map F[C][HW] (reference loss.py:67-91) as a symmetric
// rank-HW update on the fp32 matrix cores, and its backward gf (+)= D (F - mean) through the 1x1 path of
// conv_mfma2.hip.
//
// Forward: the (C/64)(C/64+1)/2 upper-triangular 64x64 tiles are split along HW over `ksplit` workgroups each
// (split-K); every workgroup writes its partial tile to a slab and a second kernel adds the slabs in index order,
// scales, and mirrors the result - a deterministic reduction without float atomics.
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>

#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));


constexpr int GT = 64;       // tile edge (channels)
#ifndef MAUA_GRAM_GK
#define MAUA_GRAM_GK 64
#endif
constexpr int GK = MAUA_GRAM_GK;  // pixels per stage
constexpr int GRS = GK + 4;  // LDS row stride (floats): 16-byte aligned rows, 4*row mod 64 banks -> conflict-free b128

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// Row means in two fixed-order stages: grid (C, splits) partial sums in fp64, then one small kernel adds each row's partials in index
// order.  A workgroup takes ~16K values of a row (rm_split: 64516 values -> 4 workgroups, 16129 and fewer -> 1): thread t adds the
// values t, t + 256, ... of its part in order, sixteen loads in flight - with 4K values per workgroup (round 4) the launch was bound by
// workgroup turnover (35840 workgroups: 37 us for NIN's six layers, 74 MB).
constexpr int RM_SPLIT = 64;  // most splits of a row (the partial sums' row length)
static __host__ __device__ inline int rm_split(int64_t HW) {
    const int64_t k = (HW + 16383) / 16384;
    return k < 1 ? 1 : k > RM_SPLIT ? RM_SPLIT : (int)k;
}
__device__ __forceinline__ double rm_part_sum(const float* __restrict__ row, int64_t HW, int split, int part, double* scratch) {
    const int64_t per = (HW + split - 1) / split, lo = part * per, hi = min(HW, lo + per);
    double acc = 0.0;
    int64_t i = lo + threadIdx.x;
    for (; i + 15 * 256 < hi; i += 16 * 256) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = row[i + k * 256];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += (double)v[k];
    }
    for (; i < hi; i += 256) acc += (double)row[i];
    return block_sum(acc, scratch);
}
__global__ void __launch_bounds__(256)
row_mean_partial_kernel(const float* __restrict__ f, double* __restrict__ partial, int64_t HW) {
    __shared__ double scratch[16];
    const double acc = rm_part_sum(f + (int64_t)blockIdx.x * HW, HW, gridDim.y, blockIdx.y, scratch);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * RM_SPLIT + blockIdx.y] = acc;
}
__global__ void row_mean_finish_kernel(const double* __restrict__ partial, float* __restrict__ mean, int C, int64_t HW) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    const int split = rm_split(HW);
    for (int k = 0; k < split; ++k) acc += partial[(int64_t)c * RM_SPLIT + k];
    mean[c] = (float)(acc / (double)HW);
}

// ... of several layers in one launch each
constexpr int RM_MAX = 8;
struct RowMeanBatch {
    const float* f[RM_MAX];
    double* partial[RM_MAX];
    float* mean[RM_MAX];
    int64_t HW[RM_MAX];
    int C[RM_MAX];
    int first[RM_MAX + 1];  // (row, part) items of the layers in front
};
__global__ void __launch_bounds__(256) row_mean_partial_batch_kernel(RowMeanBatch b) {
    __shared__ double scratch[16];
    // grid x = the (row, part) items of every layer, one layer after the other (b.first: prefix sums of C x splits); the arithmetic of
    // row_mean_partial_kernel (bit-identical means)
    int z = 0;
    while ((int)blockIdx.x >= b.first[z + 1]) ++z;
    const int64_t HW = b.HW[z];
    const int split = rm_split(HW), item = blockIdx.x - b.first[z], c = item / split, part = item - c * split;
    const double acc = rm_part_sum(b.f[z] + (int64_t)c * HW, HW, split, part, scratch);
    if (threadIdx.x == 0) b.partial[z][(int64_t)c * RM_SPLIT + part] = acc;
}
__global__ void row_mean_finish_batch_kernel(RowMeanBatch b) {
    const int z = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.C[z]) return;
    double acc = 0.0;
    const int split = rm_split(b.HW[z]);
    for (int k = 0; k < split; ++k) acc += b.partial[z][(int64_t)c * RM_SPLIT + k];
    b.mean[z][c] = (float)(acc / (double)b.HW[z]);
}

// One workgroup = one upper-triangular 64x64 tile pair (ti <= tj) x one slice of HW.  Each of the 4 waves owns a 32x32
// block.  LDS tiles are [channel][pixel] exactly as in HBM (16-byte global loads, b128 LDS writes); a lane reads 4
// consecutive pixels with one ds_read_b128 and uses them as the k-values of 4 MFMAs: half h of the wave takes pixels
// 8q+4h..8q+4h+3 of every 8-pixel group, the same pixels for the A and the B operand, so the pairing is consistent.
// On diagonal tiles the (1,0) block is the transpose of (0,1) and is not computed (gram_finish mirrors it).
__global__ void __launch_bounds__(256)
gram_partial_kernel(const float* __restrict__ f, const float* __restrict__ mean, float* __restrict__ partial, int C,
                    int64_t HW, int ksplit, int64_t chunk) {
    __shared__ __attribute__((aligned(16))) float At[GT * GRS];
    __shared__ __attribute__((aligned(16))) float Bt[GT * GRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i32 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    int pair = blockIdx.x, ti = 0;
    const int ntile = (C + GT - 1) / GT;
    while (pair >= ntile - ti) {
        pair -= ntile - ti;
        ++ti;
    }
    const int tj = ti + pair;
    const bool diag = ti == tj;
    const int ks = blockIdx.y;
    const int64_t p_begin = (int64_t)ks * chunk;
    const int64_t p_end = min(HW, p_begin + chunk);

    constexpr int NL = GT * GK / 4 / 256;  // 4 quads per tile per thread
    f32x4 ra[NL], rb[NL];
    auto load_stage = [&](int64_t p0) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = tid + 256 * q;
            const int r = e / (GK / 4), c4 = (e - r * (GK / 4)) * 4;
            const int64_t pp = p0 + c4;
            const int rowa = ti * GT + r, rowb = tj * GT + r;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            // 16-byte loads need only dword alignment in global memory, so rows of odd length (NIN's 253^2, 63^2, 31^2 planes;
            // the deep layers at 724^2) are read the same way; a row's last, partial quad goes element by element
            if (pp + 4 <= p_end) {
                if (rowa < C) {
                    va = *reinterpret_cast<const f32x4u*>(f + (int64_t)rowa * HW + pp);
                    if (mean) va -= mean[rowa];
                }
                if (!diag && rowb < C) {
                    vb = *reinterpret_cast<const f32x4u*>(f + (int64_t)rowb * HW + pp);
                    if (mean) vb -= mean[rowb];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (pp + k < p_end) {
                        if (rowa < C) va[k] = f[(int64_t)rowa * HW + pp + k] - (mean ? mean[rowa] : 0.f);
                        if (!diag && rowb < C) vb[k] = f[(int64_t)rowb * HW + pp + k] - (mean ? mean[rowb] : 0.f);
                    }
                }
            }
            ra[q] = va;
            rb[q] = vb;
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = tid + 256 * q;
            const int r = e / (GK / 4), c4 = (e - r * (GK / 4)) * 4;
            *reinterpret_cast<f32x4*>(At + r * GRS + c4) = ra[q];
            if (!diag) *reinterpret_cast<f32x4*>(Bt + r * GRS + c4) = rb[q];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool skip = diag && wi == 1 && wj == 0;  // wave-uniform

    if (p_begin < p_end) {
        load_stage(p_begin);
        for (int64_t p0 = p_begin; p0 < p_end; p0 += GK) {
            __syncthreads();  // previous stage fully consumed
            store_stage();
            __syncthreads();
            if (p0 + GK < p_end) load_stage(p0 + GK);
            if (!skip) {
                const float* at = At + (wi * 32 + i32) * GRS + 4 * half;
                const float* bt = (diag ? At : Bt) + (wj * 32 + i32) * GRS + 4 * half;
#pragma unroll
                for (int q = 0; q < GK / 8; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(at + 8 * q);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bt + 8 * q);
#pragma unroll
                    for (int cp = 0; cp < 4; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b[cp], acc, 0, 0, 0);
                }
            }
        }
    }
    float* out = partial + ((int64_t)blockIdx.x * ksplit + ks) * (GT * GT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[row * GT + wj * 32 + i32] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same partial tile on the fp16 matrix cores in fp16x3 arithmetic (conv_x3.hip: value * s = high + low fp16 part, products
// low*high, high*low, high*high on v_mfma_f32_32x32x16_f16, fp32 accumulate, un-scaled fold into an fp32 master).  The fp32
// kernel above is bound by the fp32 matrix rate (64 cycles per 32x32x2 MFMA: 80-130 TFLOP/s on every style layer, 2-3x the
// time an HBM-rate pass over the map would take); with three fp16 MFMAs per product block the arithmetic is 5x cheaper and
// the kernel becomes a streaming one, so it is built as one:
//   * loads run TWO 64-pixel stages ahead in registers (32 KB per diagonal workgroup in flight, 4 workgroups per CU);
//   * the fp16 planes [tile][part][channel][64 px] are double-buffered in LDS: ONE barrier per stage;
//   * no cross-wave exchange for the scales: a wave stages, for each tile, the unit (32-channel block w >> 1, 32-pixel half
//     w & 1), takes ITS maximum (DPP), scales it into [2^11, 2^12) and leaves the inverse scale next to the planes.  A wave's
//     32x32 output block therefore sees one scale pair per pixel half: k-steps 0-1 and 2-3 accumulate separately and are
//     folded with (inverse scale of its row block) x (inverse scale of its column block).
// Same slab format as gram_partial_kernel, same plan, same finish kernels; error against fp64 as for the fp32 kernel.
typedef _Float16 g16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g16x2 __attribute__((ext_vector_type(2)));
typedef float g32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gu32x2 __attribute__((ext_vector_type(2)));
constexpr int GXROW = 144;                       // bytes per fp16 plane row: 64 px * 2 B + 16 (conflict-free b128 reads)
constexpr int GXPLANE = GT * GXROW;              // one [channel][px] plane of one part
constexpr int GXBUF = 2 * 2 * GXPLANE + 64;      // [tile][part] planes + 8 inverse scales [tile][block][half]

__device__ __forceinline__ unsigned gx_cvt_pk(float a, float b) {
    const g32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, g16x2));
}

// (workgroup `pair_index` of the tile pairs, slab `ks`: blockIdx.x / .y of a single layer's launch, decoded by the batch kernel below)
__device__ __forceinline__ void gram_x3_partial_body(const float* __restrict__ f, const float* __restrict__ mean, float* __restrict__ partial,
                                                     int C, int64_t HW, int ksplit, int64_t chunk, int nplanes, const int pair_index,
                                                     const int ks) {
    // two buffers of `nplanes` planes (4 = [tile][part]; 2 when C <= 64: the single, diagonal tile pair needs no second tile,
    // and four such workgroups fit a CU) + 64 bytes of inverse scales each
    extern __shared__ __attribute__((aligned(16))) float smem_f32[];  // (one dynamic-LDS symbol per translation unit)
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f32);
    const int GXBUFD = nplanes * GXPLANE + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i32 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    int pair = pair_index, ti = 0;
    const int ntile = (C + GT - 1) / GT;
    while (pair >= ntile - ti) {
        pair -= ntile - ti;
        ++ti;
    }
    const int tj = ti + pair;
    const bool diag = ti == tj;
    const int64_t p_begin = (int64_t)ks * chunk;
    const int64_t p_end = min(HW, p_begin + chunk);

    // staging unit of this wave (in each tile): rows blk * 32 + lane / 8 + 8 r (r < 4), pixels pxh * 32 + (lane % 8) * 4 .. + 3
    const int blk = wave >> 1, pxh = wave & 1;
    const int srow = blk * 32 + (lane >> 3), spx = pxh * 32 + (lane & 7) * 4;
    f32x4 ra[2][4], rb[2][4];  // two stages in flight: [slot][r]
    // Raw values only: nothing here uses a loaded register, so the requests of two stages stay in flight (the row means of the
    // covariance form are subtracted when the stage is split).  Interior stages of full tiles take the branch-free path.
    auto load_unit = [&](f32x4 (&r4)[4], int tile, int64_t p0) {
        const int64_t pp = p0 + spx;
        if (p0 + GK <= p_end && tile * GT + GT <= C) {  // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r)
                r4[r] = *reinterpret_cast<const f32x4u*>(f + (int64_t)(tile * GT + srow + 8 * r) * HW + pp);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tile * GT + srow + 8 * r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < C) {
                if (pp + 4 <= p_end) {
                    v = *reinterpret_cast<const f32x4u*>(f + (int64_t)row * HW + pp);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (pp + k < p_end) v[k] = f[(int64_t)row * HW + pp + k];
                }
            }
            r4[r] = v;
        }
    };
    // row means of this lane's four rows per tile (covariance form), and which (row, pixel) cells exist at all
    float mrow[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (t == 0 ? ti : tj) * GT + srow + 8 * r;
            mrow[t][r] = (mean && row < C) ? mean[row] : 0.f;
        }
    auto load_stage = [&](int slot, int64_t p0) {
        if (slot == 0) {
            load_unit(ra[0], ti, p0);
            if (!diag) load_unit(rb[0], tj, p0);
        } else {
            load_unit(ra[1], ti, p0);
            if (!diag) load_unit(rb[1], tj, p0);
        }
    };
    // split one unit (scale from this wave's own maximum) and write it into buffer `buf`
    auto store_unit = [&](f32x4 (&r4)[4], int tile_slot, int64_t p0, unsigned char* buf) {
        if (mean) {  // covariance form: centre the cells that exist (padding cells stay zero)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (tile_slot == 0 ? ti : tj) * GT + srow + 8 * r;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (row < C && p0 + spx + k < p_end) r4[r][k] -= mrow[tile_slot][r];
            }
        }
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(fmaxf(fabsf(r4[r][0]), fabsf(r4[r][1])), fmaxf(fabsf(r4[r][2]), fabsf(r4[r][3]))));
        m = wave_max_nonneg(m);
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
        e = m > 0.f ? max(e, -100) : 11;
        const float sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
        if (lane == 0)
            reinterpret_cast<float*>(buf + nplanes * GXPLANE)[tile_slot * 4 + blk * 2 + pxh] = __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v0 = r4[r][0] * sx, v1 = r4[r][1] * sx, v2 = r4[r][2] * sx, v3 = r4[r][3] * sx;
            const unsigned h0 = gx_cvt_pk(v0, v1), h1 = gx_cvt_pk(v2, v3);
            const g16x2 hh0 = __builtin_bit_cast(g16x2, h0), hh1 = __builtin_bit_cast(g16x2, h1);
            const unsigned l0 = gx_cvt_pk(v0 - (float)hh0[0], v1 - (float)hh0[1]), l1 = gx_cvt_pk(v2 - (float)hh1[0], v3 - (float)hh1[1]);
            unsigned char* dst = buf + (tile_slot * 2) * GXPLANE + (srow + 8 * r) * GXROW + spx * 2;
            *reinterpret_cast<gu32x2*>(dst) = gu32x2{h0, h1};
            *reinterpret_cast<gu32x2*>(dst + GXPLANE) = gu32x2{l0, l1};
        }
    };
    auto store_stage = [&](int slot, int64_t p0, unsigned char* buf) {
        if (slot == 0) {
            store_unit(ra[0], 0, p0, buf);
            if (!diag) store_unit(rb[0], 1, p0, buf);
        } else {
            store_unit(ra[1], 0, p0, buf);
            if (!diag) store_unit(rb[1], 1, p0, buf);
        }
    };

    f32x16 master, acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) master[r] = acc0[r] = acc1[r] = 0.f;
    const bool skip = diag && wi == 1 && wj == 0;  // wave-uniform: the mirrored block of a diagonal tile
    const int bslot = diag ? 0 : 1;                 // the column operand comes from tile tj's planes
    const int a_off = (wi * 32 + i32) * GXROW + half * 16;
    const int b_off = (bslot * 2) * GXPLANE + (wj * 32 + i32) * GXROW + half * 16;

    const int64_t nstages = p_begin < p_end ? (p_end - p_begin + GK - 1) / GK : 0;
    if (nstages > 0) {
        load_stage(0, p_begin);
        if (nstages > 1) load_stage(1, p_begin + GK);
        store_stage(0, p_begin, smem);
        if (nstages > 2) load_stage(0, p_begin + 2 * GK);
        __syncthreads();
        for (int64_t st = 0; st < nstages; ++st) {
            unsigned char* cur = smem + (st & 1) * GXBUFD;
            unsigned char* nxt = smem + ((st + 1) & 1) * GXBUFD;
            if (!skip) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const g16x8 ah = *reinterpret_cast<const g16x8*>(cur + a_off + q * 32);
                    const g16x8 al = *reinterpret_cast<const g16x8*>(cur + GXPLANE + a_off + q * 32);
                    const g16x8 bh = *reinterpret_cast<const g16x8*>(cur + b_off + q * 32);
                    const g16x8 bl = *reinterpret_cast<const g16x8*>(cur + GXPLANE + b_off + q * 32);
                    if (q < 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                    } else {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                    }
                }
            }
            // stage st + 1 (loaded a stage ago) -> the other buffer; then request stage st + 3 into the freed registers
            if (st + 1 < nstages) {
                if ((st + 1) & 1) store_stage(1, p_begin + (st + 1) * GK, nxt);
                else store_stage(0, p_begin + (st + 1) * GK, nxt);
                if (st + 3 < nstages) {
                    if ((st + 1) & 1) load_stage(1, p_begin + (st + 3) * GK);
                    else load_stage(0, p_begin + (st + 3) * GK);
                }
            }
            if (!skip) {
                const float* inv = reinterpret_cast<const float*>(cur + nplanes * GXPLANE);
                const float s0 = inv[wi * 2 + 0] * inv[bslot * 4 + wj * 2 + 0], s1 = inv[wi * 2 + 1] * inv[bslot * 4 + wj * 2 + 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[r] = fmaf(acc0[r], s0, fmaf(acc1[r], s1, master[r]));
                    acc0[r] = 0.f;
                    acc1[r] = 0.f;
                }
            }
            __syncthreads();  // buffer `nxt` is complete; every wave is done with `cur` (rewritten two stages on)
        }
    }
    float* out = partial + ((int64_t)pair_index * ksplit + ks) * (GT * GT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[row * GT + wj * 32 + i32] = master[r];
    }
}

// The same partial product in 128 x 128 blocks, for layers of 128 channels and more: a workgroup = a PAIR of
// 128-channel tiles x one slice of HW.  A 64 x 64 workgroup splits every value it stages once for 12 MFMAs per wave and stage (~300
// vector-ALU / LDS instructions: issue-bound, profiles/probes_r03.md section 6); here a stage's values are split once for 48 MFMAs per
// multiplying wave - half the split and LDS-store work per product, 0.67 fragment reads per MFMA instead of 1.33.
//
// Eight waves, ONE workgroup per CU, two ROLES (one wave of each per SIMD):
//   waves 4 - 7 stage: wave w loads rows 32 w ... 32 w + 31 of both tiles, 64 pixels per stage, two stages of loads in flight in its
//     registers; one power-of-two scale per (tile, its 32 rows, stage) from their maximum; the two fp16 parts go into the planes of the
//     stage's LDS buffer ([tile][part][128 rows][64 px] = 72 KB, two buffers), the inverse scale beside them;
//   waves 0 - 3 multiply: wave (wi, wj) owns the 64 x 64 sub-block = 2 x 2 accumulators of 32 x 32, folds them into fp32 masters with the
//     product of the row-block and column-block inverse scales once per stage, and writes the slab of the 64-channel tile pair at the end.
// ONE barrier per stage: behind it buffer st is complete and buffer st + 1 is free (the multiplying waves arrive with stage st - 1 done).
// Every multiplying wave runs the same 48 MFMAs per stage - the mirrored sub-block of a diagonal pair is computed and not stored.
// Same slabs (per 64 x 64 tile pair, in gram_partial_kernel's format), same finishing kernels.
//
// Why not four waves that do both, two workgroups per CU (the first form: 1.3 - 1.9 x the 64 x 64 kernel): with two such waves on a
// SIMD a slab came out now and then with ONE accumulator register of a wave's (1, 1) block zero in lanes 48 - 63 - 16 zeros in a 64 x 64
// block, a different block from launch to launch - and never with one; not explained (profiles/probes_r04.md section 2 has the
// experiments, tools/stress_gram.py is the test that sees it).  Here a SIMD has one wave that multiplies and one that does not.
constexpr int GX128_PLANE = 128 * GXROW;             // one [row][px] plane of one part of one 128-channel tile
constexpr int GX128_BUF = 2 * 2 * GX128_PLANE + 64;  // [tile][part] planes + inverse scales [tile][32-row block]
constexpr int GX128_LDS = 2 * GX128_BUF;             // two stages: 144 KB

__device__ __forceinline__ void gram_x3_partial128_body(const float* __restrict__ f, const float* __restrict__ mean, float* __restrict__ partial,
                                                        int C, int64_t HW, int ksplit, int64_t chunk, const int pair_index, const int ks) {
    extern __shared__ __attribute__((aligned(16))) float smem_f32[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f32);
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const int ntile128 = (C + 127) / 128, ntile64 = (C + GT - 1) / GT;
    int pair = pair_index, Ti = 0;
    while (pair >= ntile128 - Ti) {
        pair -= ntile128 - Ti;
        ++Ti;
    }
    const int Tj = Ti + pair;
    const bool diag = Ti == Tj;
    const int64_t p_begin = (int64_t)ks * chunk;
    const int64_t p_end = min(HW, p_begin + chunk);
    const int nstages = p_begin < p_end ? (int)((p_end - p_begin + GK - 1) / GK) : 0;
    const int inv_off = 4 * GX128_PLANE;  // (the scales of a buffer: behind its four planes, whatever the pair uses of them)

    if (wave8 >= 4) {
        // ---------------------------------------------------------------- staging waves
        // units of this wave (in each tile): rows wave * 32 + lane / 8 + 8 r (r < 4), pixels pxh * 32 + (lane % 8) * 4 .. + 3, pxh = 0, 1
        const int srow = wave * 32 + (lane >> 3), spx = (lane & 7) * 4;
        f32x4 rr[2][2][2][4];  // [stage parity][tile i / j][pixel half][r]: two stages in flight
        // Loads through a buffer descriptor over the whole map (32-bit offsets: the host side sends maps of 2^29 values and more to the 64 x 64
        // kernel): rows beyond C are beyond its range and come back as zeros; pixels beyond the slice are zeroed in the (wave-uniform) tail path.
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f), 0, (unsigned)((int64_t)C * HW * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mean ? mean : f), 0, mean ? (unsigned)C * 4u : 0u, 0x00020000);
        const unsigned vbase = ((unsigned)srow * (unsigned)HW + (unsigned)spx) * 4u;
        auto load_unit = [&](f32x4 (&r4)[4], int tile, int64_t p0) {
            unsigned vb = vbase;
            asm volatile("" : "+v"(vb));  // (a unit's offset = one per-thread register + scalars, added per load)
            const unsigned sb = ((unsigned)(tile * 128) * (unsigned)HW + (unsigned)p0) * 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                r4[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, vb + (sb + (unsigned)(8 * r) * (unsigned)HW * 4u), 0, 0));
        };
        auto load_stage = [&](f32x4 (&r)[2][2][4], int64_t p0) {
            load_unit(r[0][0], Ti, p0);
            load_unit(r[0][1], Ti, p0 + 32);
            if (!diag) {
                load_unit(r[1][0], Tj, p0);
                load_unit(r[1][1], Tj, p0 + 32);
            }
        };
        // split the wave's 32 rows x 64 pixels of one tile (one scale from their maximum) and write them into the planes of `buf`
        auto store_tile = [&](f32x4 (&r4)[2][4], int tile_slot, int64_t p0, unsigned char* buf) {
            if (p0 + GK > p_end) {  // wave-uniform: the slice's last, partial stage - cells beyond it are zeros
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int left = (int)(p_end - p0) - 32 * h - spx;  // cells of this thread's four that exist (<= 0: none)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k >= left) r4[h][r][k] = 0.f;
                }
            }
            if (mean) {  // covariance form: centre the cells that exist (padding cells stay zero; rows beyond C read a mean of 0)
                unsigned vm = (unsigned)srow * 4u;
                asm volatile("" : "+v"(vm));
                const unsigned sm = (unsigned)((tile_slot == 0 ? Ti : Tj) * 128) * 4u;
                const bool tail = p0 + GK > p_end;  // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, vm + (sm + 32u * r), 0, 0));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (tail) {
                            const int left = (int)(p_end - p0) - 32 * h - spx;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < left) r4[h][r][k] -= mr;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) r4[h][r][k] -= mr;
                        }
                    }
                }
            }
            float m = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    m = fmaxf(m, fmaxf(fmaxf(fabsf(r4[h][r][0]), fabsf(r4[h][r][1])), fmaxf(fabsf(r4[h][r][2]), fabsf(r4[h][r][3]))));
            m = wave_max_nonneg(m);
            int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
            e = m > 0.f ? max(e, -100) : 11;
            const float sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
            if (lane == 0) reinterpret_cast<float*>(buf + inv_off)[tile_slot * 4 + wave] = __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = r4[h][r][0] * sx, v1 = r4[h][r][1] * sx, v2 = r4[h][r][2] * sx, v3 = r4[h][r][3] * sx;
                    const unsigned h0 = gx_cvt_pk(v0, v1), h1 = gx_cvt_pk(v2, v3);
                    const g16x2 hh0 = __builtin_bit_cast(g16x2, h0), hh1 = __builtin_bit_cast(g16x2, h1);
                    const unsigned l0 = gx_cvt_pk(v0 - (float)hh0[0], v1 - (float)hh0[1]), l1 = gx_cvt_pk(v2 - (float)hh1[0], v3 - (float)hh1[1]);
                    unsigned char* dst = buf + (tile_slot * 2) * GX128_PLANE + (srow + 8 * r) * GXROW + (32 * h + spx) * 2;
                    *reinterpret_cast<gu32x2*>(dst) = gu32x2{h0, h1};
                    *reinterpret_cast<gu32x2*>(dst + GX128_PLANE) = gu32x2{l0, l1};
                }
        };
        if (nstages > 0) load_stage(rr[0], p_begin);
        if (nstages > 1) load_stage(rr[1], p_begin + GK);
        // (two stages per trip: the register set of a stage is a compile-time index)
        for (int st = 0; st < nstages; st += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int s1 = st + par;
                if (s1 < nstages) {  // wave-uniform
                    unsigned char* buf = smem + par * GX128_BUF;
                    const int64_t p0 = p_begin + (int64_t)s1 * GK;
                    store_tile(rr[par][0], 0, p0, buf);
                    if (!diag) store_tile(rr[par][1], 1, p0, buf);
                    if (s1 + 2 < nstages) load_stage(rr[par], p0 + 2 * GK);
                    __syncthreads();  // buffer `par` holds stage s1; the multiplying waves are done with the other one
                }
            }
        }
        return;
    }
    // -------------------------------------------------------------------- multiplying waves
    const int i32 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 master[2][2], acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) master[a][b][r] = acc[a][b][r] = 0.f;
    const int bslot = diag ? 0 : 1;
    const int a_off = (wi * 64 + i32) * GXROW + half * 16;
    const int b_off = (bslot * 2) * GX128_PLANE + (wj * 64 + i32) * GXROW + half * 16;
    for (int st = 0; st < nstages; ++st) {
        __syncthreads();  // stage st is in its buffer
        const unsigned char* buf = smem + (st & 1) * GX128_BUF;
        const float* inv_lds = reinterpret_cast<const float*>(buf + inv_off);
        float s_row[2], s_col[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            s_row[a] = inv_lds[2 * wi + a];
            s_col[a] = inv_lds[bslot * 4 + 2 * wj + a];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            g16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = *reinterpret_cast<const g16x8*>(buf + a_off + a * 32 * GXROW + q * 32);
                al[a] = *reinterpret_cast<const g16x8*>(buf + GX128_PLANE + a_off + a * 32 * GXROW + q * 32);
                bh[a] = *reinterpret_cast<const g16x8*>(buf + b_off + a * 32 * GXROW + q * 32);
                bl[a] = *reinterpret_cast<const g16x8*>(buf + GX128_PLANE + b_off + a * 32 * GXROW + q * 32);
            }
            // (the four accumulators in turn: no MFMA waits for the one before it)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float sc = s_row[a] * s_col[b];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[a][b][r] = fmaf(acc[a][b][r], sc, master[a][b][r]);
                    acc[a][b][r] = 0.f;
                }
            }
    }
    if (nstages == 0) return;
    // the wave's 64 x 64 sub-block is the slab of the 64-channel tile pair (2 Ti + wi, 2 Tj + wj); the mirrored one of a diagonal pair is not stored
    const int ti64 = 2 * Ti + wi, tj64 = 2 * Tj + wj;
    if (ti64 > tj64 || tj64 >= ntile64) return;
    const int p64 = ti64 * ntile64 - ti64 * (ti64 - 1) / 2 + (tj64 - ti64);
    float* out = partial + ((int64_t)p64 * ksplit + ks) * (GT * GT) + half * 4 * GT + i32;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(a * 32 + (r & 3) + 8 * (r >> 2)) * GT + b * 32] = master[a][b][r];
}

// Workgroup -> (tile pair, slice) of a layer that multiplies in 128 x 128 blocks.  The pairs of ONE slice read the same 128-channel tiles
// of that slice (C = 512: ten pairs over four tiles - every tile four times); dealt out pair-fastest, consecutive workgroups land on
// eight different XCDs and every one of those reads comes from memory (L2 hit rate 0.11, 436 MB of tiles read for 243 MB of maps per
// 1024 x 1024 iteration: profiles/pmc_r04_traffic.json).  So: XCD x (workgroups x, x + 8, ... of the layer's range, which starts at a
// multiple of 8) owns the slices [x spx, (x + 1) spx) and walks them pair by pair - the pairs of a slice run on the CUs of one L2 at about
// the same time.  `local` = the workgroup's number inside the layer's padded range of 8 spx npairs; false: nothing to do.  Same slabs,
// same sums: only the placement changes.
__device__ __forceinline__ bool gram128_item(int local, int npairs, int ksplit, int* pair, int* ks) {
    const int spx = (ksplit + 7) >> 3;
    const int j = local >> 3;
    const int slice = (local & 7) * spx + j / npairs;
    *pair = j % npairs;
    *ks = slice;
    return j < spx * npairs && slice < ksplit;
}
__host__ __device__ inline int gram128_grid(int npairs, int ksplit) { return 8 * ((ksplit + 7) / 8) * npairs; }

__global__ void __launch_bounds__(512, 1)
gram_x3_partial128_kernel(const float* __restrict__ f, const float* __restrict__ mean, float* __restrict__ partial, int C,
                          int64_t HW, int ksplit, int64_t chunk, int npairs) {
    int pair, ks;
    if (!gram128_item((int)blockIdx.x, npairs, ksplit, &pair, &ks)) return;
    gram_x3_partial128_body(f, mean, partial, C, HW, ksplit, chunk, pair, ks);
}

__global__ void __launch_bounds__(256, 2)
gram_x3_partial_kernel(const float* __restrict__ f, const float* __restrict__ mean, float* __restrict__ partial, int C,
                       int64_t HW, int ksplit, int64_t chunk, int nplanes) {
    gram_x3_partial_body(f, mean, partial, C, HW, ksplit, chunk, nplanes, blockIdx.x, blockIdx.y);
}

// The partial kernels of up to GB_MAX style layers in ONE launch (grid z = layer, x / y = the largest layer's; a layer's surplus
// workgroups leave at once): below 768 x 768 the Gram chains run in stream order and each layer's launch is 10 - 25 us of mostly latency
// on a part of the chip; together they fill it.  Every layer keeps its own plan (tile pairs, slabs): the slabs hold the same bits.
constexpr int GB_MAX = 8;
struct GramPartialBatch {
    const float* f[GB_MAX];
    const float* mean[GB_MAX];  // row means (covariance form) or null
    float* partial[GB_MAX];
    int64_t HW[GB_MAX], chunk[GB_MAX];
    int C[GB_MAX], ksplit[GB_MAX], npairs[GB_MAX], nplanes[GB_MAX];  // npairs: workgroups per slab (64- or 128-channel tile pairs)
    int first[GB_MAX + 1];  // workgroups [first[z], first[z + 1]) belong to layer z: a dense 1-D grid (no workgroup is launched to leave)
    int count;
};
__global__ void __launch_bounds__(256, 2) gram_x3_partial_batch_kernel(GramPartialBatch b) {
    int z = 0;
    while (z + 1 < b.count && (int)blockIdx.x >= b.first[z + 1]) ++z;
    const int local = blockIdx.x - b.first[z];
    const int ks = local / b.npairs[z];  // (tile pair fastest, as blockIdx.x of the layer's own launch)
    gram_x3_partial_body(b.f[z], b.mean[z], b.partial[z], b.C[z], b.HW[z], b.ksplit[z], b.chunk[z], b.nplanes[z], local - ks * b.npairs[z], ks);
}
// ... and the layers that multiply in 128 x 128 blocks: a launch of their own (one workgroup per CU, see gram_x3_partial128_body)
__global__ void __launch_bounds__(512, 1) gram_x3_partial128_batch_kernel(GramPartialBatch b) {
    int z = 0;
    while (z + 1 < b.count && (int)blockIdx.x >= b.first[z + 1]) ++z;
    int pair, ks;
    if (!gram128_item((int)blockIdx.x - b.first[z], b.npairs[z], b.ksplit[z], &pair, &ks)) return;  // (b.first: multiples of 8)
    gram_x3_partial128_body(b.f[z], b.mean[z], b.partial[z], b.C[z], b.HW[z], b.ksplit[z], b.chunk[z], pair, ks);
}

// sum of slab values base[k * stride_elems] for k = 0, kstride, 2 kstride, ... < ksplit in index order (fp64): eight loads in flight per
// step - the finishing passes are a handful of workgroups whose time is the latency of this chain - and the additions in the order of a
// plain loop (same bits).
__device__ __forceinline__ double slab_sum(const float* __restrict__ base, int ksplit, int kstride) {
    double sd = 0.0;
    int k = 0;
    for (; k + 7 * kstride < ksplit; k += 8 * kstride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(k + u * kstride) * (GT * GT)];
#pragma unroll
        for (int u = 0; u < 8; ++u) sd += (double)v[u];
    }
    for (; k < ksplit; k += kstride) sd += (double)base[(int64_t)k * (GT * GT)];
    return sd;
}

// First level of the slab sum when there are many slabs (relu1_1 at 1024x1024: 745): grid (pairs, 16, groups), every thread
// adds the GF_FOLD slabs of its group in index order (fp64) and leaves the sum in the group's first slab.  Fixed order.
constexpr int GF_FOLD = 32;
__global__ void __launch_bounds__(256)
gram_fold_kernel(float* __restrict__ partial, int ksplit) {
    float* base = partial + (int64_t)blockIdx.x * ksplit * (GT * GT) + blockIdx.y * 256 + threadIdx.x;
    const int k0 = blockIdx.z * GF_FOLD, k1 = min(k0 + GF_FOLD, ksplit);
    const double sd = slab_sum(base + (int64_t)k0 * (GT * GT), k1 - k0, 1);
    base[(int64_t)k0 * (GT * GT)] = (float)sd;
}

// gram_fold_kernel for the layers of a batch: grid z = layer * groups of the layer with the most + group
struct GramFoldBatch {
    float* partial[GB_MAX];
    int ksplit[GB_MAX], npairs[GB_MAX];
    int max_groups;
};
__global__ void __launch_bounds__(256) gram_fold_batch_kernel(GramFoldBatch b) {
    const int z = blockIdx.z / b.max_groups, grp = blockIdx.z - z * b.max_groups;
    const int ksplit = b.ksplit[z];
    const int k0 = grp * GF_FOLD, k1 = min(k0 + GF_FOLD, ksplit);
    if ((int)blockIdx.x >= b.npairs[z] || ksplit <= 2 * GF_FOLD || k0 >= ksplit) return;
    float* base = b.partial[z] + (int64_t)blockIdx.x * ksplit * (GT * GT) + blockIdx.y * 256 + threadIdx.x;
    const double sd = slab_sum(base + (int64_t)k0 * (GT * GT), k1 - k0, 1);
    base[(int64_t)k0 * (GT * GT)] = (float)sd;
}

__global__ void __launch_bounds__(256)
gram_finish_kernel(const float* __restrict__ partial, float* __restrict__ gram, int C, int ksplit, int kstride, float scale) {
    // grid = (tile pairs, 16): every thread owns ONE element of the 64x64 tile and adds its `ksplit` slab values in
    // index order (fixed order -> deterministic); consecutive threads read consecutive addresses of each slab.
    int pair = blockIdx.x, ti = 0;
    const int ntile = (C + GT - 1) / GT;
    while (pair >= ntile - ti) {
        pair -= ntile - ti;
        ++ti;
    }
    const int tj = ti + pair;
    const float* base = partial + (int64_t)blockIdx.x * ksplit * (GT * GT);
    const int e = blockIdx.y * 256 + threadIdx.x;
    int er = e / GT, ec = e % GT;
    // diagonal tiles: the lower triangle is read from the upper one - block (1,0) was not computed at all, and G comes out
    // exactly symmetric whatever arithmetic produced the slabs
    const int src = (ti == tj && er > ec) ? ec * GT + er : e;
    const double sd = slab_sum(base + src, ksplit, kstride);  // fp64 keeps the split-K sum exact to fp32 rounding; slabs k = 0, kstride, 2 kstride, ...
    const float s = (float)(sd * (double)scale);
    const int gi = ti * GT + er, gj = tj * GT + ec;
    if (gi < C && gj < C) {
        gram[(int64_t)gi * C + gj] = s;
        if (ti != tj) gram[(int64_t)gj * C + gi] = s;
    }
}

// gram_finish_kernel and the MSE against the style target in one launch (loss.py:150-157 on top of GramMatrix.forward): G as
// above, D = grad_scale (G - T) (mirrored like G), and the workgroup's share of sum (G - T)^2 - off-diagonal tiles count twice -
// as one partial sum of the slot's ledger record.
__global__ void __launch_bounds__(256)
gram_finish_mse_kernel(const float* __restrict__ partial, float* __restrict__ gram, int C, int ksplit, int kstride, float scale,
                       const float* __restrict__ target, float* __restrict__ dmat, float grad_scale, float loss_scale,
                       double* __restrict__ rec) {
    __shared__ double scratch[16];
    int pair = blockIdx.x, ti = 0;
    const int ntile = (C + GT - 1) / GT;
    while (pair >= ntile - ti) {
        pair -= ntile - ti;
        ++ti;
    }
    const int tj = ti + pair;
    const float* base = partial + (int64_t)blockIdx.x * ksplit * (GT * GT);
    const int e = blockIdx.y * 256 + threadIdx.x;
    const int er = e / GT, ec = e % GT;
    const int src = (ti == tj && er > ec) ? ec * GT + er : e;
    const double sd = slab_sum(base + src, ksplit, kstride);
    const float s = (float)(sd * (double)scale);
    const int gi = ti * GT + er, gj = tj * GT + ec;
    double sq = 0.0;
    if (gi < C && gj < C) {
        const float d = s - target[(int64_t)gi * C + gj];
        gram[(int64_t)gi * C + gj] = s;
        dmat[(int64_t)gi * C + gj] = grad_scale * d;
        sq = (double)d * (double)d;
        if (ti != tj) {
            const float dt = s - target[(int64_t)gj * C + gi];
            gram[(int64_t)gj * C + gi] = s;
            dmat[(int64_t)gj * C + gi] = grad_scale * dt;
            sq += (double)dt * (double)dt;
        }
    }
    sq = block_sum(sq, scratch);
    if (threadIdx.x == 0) {
        rec[2 + blockIdx.x * gridDim.y + blockIdx.y] = sq;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            rec[0] = (double)(gridDim.x * gridDim.y);
            rec[1] = (double)loss_scale;
        }
    }
}

// gram_finish_mse_kernel for up to GB_MAX style layers in ONE launch (grid z = layer; a layer's surplus workgroups leave at once):
// the Gram / loss chains of an evaluation need their D matrices only when the backward pass starts, so their finishing passes - each a
// latency-bound launch of a few microseconds of work - can all wait for the last partial kernel.  Same arithmetic, same summation
// order, same ledger records as the per-layer launch: bit-identical results.
struct GramFinishBatch {
    const float* partial[GB_MAX];
    float* gram[GB_MAX];
    const float* target[GB_MAX];
    float* dmat[GB_MAX];
    double* rec[GB_MAX];
    int C[GB_MAX], ksplit[GB_MAX], kstride[GB_MAX], npairs[GB_MAX];
    float scale[GB_MAX], grad_scale[GB_MAX], loss_scale[GB_MAX];
};
__global__ void __launch_bounds__(256) gram_finish_mse_batch_kernel(GramFinishBatch b) {
    __shared__ double scratch[16];
    const int z = blockIdx.z;
    if ((int)blockIdx.x >= b.npairs[z]) return;  // (whole workgroup)
    const int C = b.C[z], ksplit = b.ksplit[z], kstride = b.kstride[z];
    int pair = blockIdx.x, ti = 0;
    const int ntile = (C + GT - 1) / GT;
    while (pair >= ntile - ti) {
        pair -= ntile - ti;
        ++ti;
    }
    const int tj = ti + pair;
    const float* base = b.partial[z] + (int64_t)blockIdx.x * ksplit * (GT * GT);
    const int e = blockIdx.y * 256 + threadIdx.x;
    const int er = e / GT, ec = e % GT;
    const int src = (ti == tj && er > ec) ? ec * GT + er : e;
    const double sd = slab_sum(base + src, ksplit, kstride);
    const float s = (float)(sd * (double)b.scale[z]);
    const int gi = ti * GT + er, gj = tj * GT + ec;
    const float* __restrict__ target = b.target[z];
    float* __restrict__ gram = b.gram[z];
    float* __restrict__ dmat = b.dmat[z];
    const float grad_scale = b.grad_scale[z];
    double sq = 0.0;
    if (gi < C && gj < C) {
        const float d = s - target[(int64_t)gi * C + gj];
        gram[(int64_t)gi * C + gj] = s;
        dmat[(int64_t)gi * C + gj] = grad_scale * d;
        sq = (double)d * (double)d;
        if (ti != tj) {
            const float dt = s - target[(int64_t)gj * C + gi];
            gram[(int64_t)gj * C + gi] = s;
            dmat[(int64_t)gj * C + gi] = grad_scale * dt;
            sq += (double)dt * (double)dt;
        }
    }
    sq = block_sum(sq, scratch);
    if (threadIdx.x == 0) {
        double* rec = b.rec[z];
        rec[2 + blockIdx.x * gridDim.y + blockIdx.y] = sq;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            rec[0] = (double)(b.npairs[z] * gridDim.y);
            rec[1] = (double)b.loss_scale[z];
        }
    }
}

// bias[c] = - sum_k D[k][c] * mean[k]   (the centering term of gf = D (F - mean 1^T)).  D is symmetric, so row c is
// read instead of column c: one wave per output, coalesced loads, fixed-order wave reduction.
__global__ void __launch_bounds__(256) center_bias_kernel(const float* __restrict__ d, const float* __restrict__ mean,
                                                          float* __restrict__ bias, int C) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float* row = d + (int64_t)c * C;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s = fmaf(row[k], mean[k], s);
    s = wave_sum(s);
    if (lane == 0) bias[c] = -s;
}

// Gram backward gf[c][p] (+)= sum_k D[k][c] * (F[k][p] - mean[k]), masked by relu_mask.  Bandwidth-bound for the
// shallow layers (C = 64: three passes over a 268 MB map at 1024x1024), so the kernel is built around wide, few
// accesses: 32 input channels per LDS stage (64 MFMAs per barrier and wave), F staged with 16-byte loads in its native
// [channel][pixel] order (the B operand then reads consecutive pixels with ds_read_b32, conflict-free), D rows likewise.
// Workgroup = 64 output channels x 256 pixels, wave = 64 x 64 (2 x 2 accumulators), double-buffered LDS (80 KB).
constexpr int GB_KC = 32, GB_PX = 256, GB_CO = 64;

template <bool ACC, bool MASK>
__global__ void __launch_bounds__(256)
gram_bwd_kernel(const float* __restrict__ d, const float* __restrict__ f, const float* __restrict__ mean,
                const float* __restrict__ rmask, float* __restrict__ gf, int C, int64_t HW, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem_f32[];  // 2 x (Xl[32][256] + Wl[32][64])
    float* smem = smem_f32;
    constexpr int BUF = GB_KC * GB_PX + GB_KC * GB_CO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * GB_PX;
    const int co0 = blockIdx.y * GB_CO;

    f32x4 rx[8], rw[2];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;           // quad index: 32 rows x 64 quads
            const int r = e >> 6, q = e & 63;
            const int64_t pp = p0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + r < C && pp < HW) {
                v = *reinterpret_cast<const f32x4*>(f + (int64_t)(k0 + r) * HW + pp);
                if (mean) v -= mean[k0 + r];
            }
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;           // 32 rows x 16 quads
            const int r = e >> 4, q = e & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + r < C && co0 + 4 * q < C) v = *reinterpret_cast<const f32x4*>(d + (int64_t)(k0 + r) * C + co0 + 4 * q);
            rw[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
        float* xl = smem + buf * BUF;
        float* wl = xl + GB_KC * GB_PX;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(xl + (tid + 256 * i) * 4) = rx[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(wl + (tid + 256 * i) * 4) = rw[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nst = (C + GB_KC - 1) / GB_KC;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) load_stage((st + 1) * GB_KC);
        const float* xl = smem + cur * BUF;
        const float* wl = xl + GB_KC * GB_PX;
#pragma unroll
        for (int kp = 0; kp < GB_KC / 2; ++kp) {
            const int kk = 2 * kp + half;
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = wl[kk * GB_CO + t * 32 + j];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = xl[kk * GB_PX + (wave * 2 + u) * 32 + j];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (st + 1 < nst) store_stage(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t pix = p0 + (wave * 2 + u) * 32 + j;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // read-modify-write in two sweeps: all loads of the 16 registers first, then all stores - otherwise every
            // load waits behind the previous store (same array) and the epilogue becomes 64 serial round trips
            float prev[16], msk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool ok = pix < HW && co < C;
                const int64_t o = ok ? (int64_t)co * HW + pix : 0;
                // compile-time switches: with run-time flags hipcc branches around every load and waits for each one
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = gf[o];
                if constexpr (MASK) msk[r] = rmask[o];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pix < HW && co < C) {
                    const float v = acc[t][u][r] + prev[r];
                    gf[(int64_t)co * HW + pix] = msk[r] > 0.f ? v : 0.f;
                }
            }
        }
    }
}

// Layers of 128 channels and more multiply in 128 x 128 blocks (gram_x3_partial128_kernel, fp16x3 route only) on maps its 32-bit buffer
// offsets reach (row index up to c + 127), of 16 stages and more (MAUA_GRAM_T128_MIN_HW pixels; below, the 64 x 64 tiles' four times as many
// workgroups fill the chip better: 512 x 256 10.0 us against 8.8) and - by default - a whole number of stages.  That last condition is a
// measured one (profiles/probes_r04.md section 2): the 128 x 128 launch is one more launch beside the 64 x 64 one of the layers that stay
// there, and where not every layer qualifies or the gain per layer is small that costs what the blocks win - VGG-19 at 362 / 724 / 1448
// (-1.1 % ... +0.4 % of a step) and NIN's 127 x 127 / 63 x 63 maps (-0.9 %) against +1.2 % / +1.0 % at 1024 / 2048.
// MAUA_GRAM_T128 (read once per process): 0 = 64 x 64 everywhere, 2 = every map of MIN_HW pixels and more (ragged last stages included).
static bool gram_tile128(int c, int64_t hw) {
    const int t128 = (int)tuning("gram_t128", 1);
    const int mode = tuning("gram_x3", 1) == 0 ? 0 : (t128 >= 0 && t128 <= 2 ? t128 : 1);
    const int64_t min_hw = (int64_t)tuning("gram_t128_min_hw", 16 * GK);
    return mode && c >= 128 && hw >= min_hw && (mode == 2 || hw % GK == 0) && (int64_t)(c + 128) * hw < (1ll << 29);
}
// npairs: 64-channel tile pairs = slabs per slice of HW (the slab format of every kernel); *wgs_per_slab (nullable): workgroups per slice
static void gram_plan(int c, int64_t hw, int* npairs, int* ksplit, int64_t* chunk, int* wgs_per_slab = nullptr) {
    const int ntile = (c + GT - 1) / GT;
    *npairs = ntile * (ntile + 1) / 2;
    const int nt128 = (c + 127) / 128;
    const int wgs = gram_tile128(c, hw) ? nt128 * (nt128 + 1) / 2 : *npairs;
    if (wgs_per_slab) *wgs_per_slab = wgs;
    const int64_t stages = (hw + GK - 1) / GK;
    int64_t want = (768 + wgs - 1) / wgs;
    if (gram_tile128(c, hw)) {
        // one workgroup per CU, and the pairs of a slice on the CUs of ONE XCD (gram128_item): spx = 32 / pairs slices per XCD fill its 32 CUs
        // once; one such round (8 spx slices) or two, whichever is nearer to slices of 16 stages (fewer, longer slices = fewer slabs for the
        // finishing kernels: 512 workgroups per layer at 1024 x 1024 cost them 35 us instead of 23, and the partial launch 120 instead of 112)
        const int64_t spx = wgs >= 32 ? 1 : 32 / wgs;
        want = stages / 16 >= 12 * spx ? 16 * spx : 8 * spx;
    }
    if (want > stages) want = stages;
    if (want < 1) want = 1;
    int64_t ch = ((hw + want - 1) / want + GK - 1) / GK * GK;
    *chunk = ch;
    *ksplit = (int)((hw + ch - 1) / ch);
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_gram_block(int c, int64_t hw) { return c > 0 && hw > 0 && gram_tile128(c, hw) ? 128 : 64; }

size_t maua_gram_workspace_bytes(int c, int64_t hw) {
    if (c <= 0 || hw <= 0 || c > (1 << 16) || hw >= (1ll << 30)) return 0;
    int npairs, ksplit;
    int64_t chunk;
    gram_plan(c, hw, &npairs, &ksplit, &chunk);
    const size_t slabs = (size_t)npairs * ksplit * GT * GT * sizeof(float) + (size_t)((c + 63) / 64 * 64) * sizeof(float);
    const size_t row_partials = (size_t)c * RM_SPLIT * sizeof(double);  // covariance form: reused before the slabs
    const size_t bwd_split = hw < (1ll << 30) ? conv1x1_x3_workspace(1, c, hw, c) : 0;  // maua_gram_bwd's split channel loop
    size_t need = slabs > row_partials ? slabs : row_partials;
    return need > bwd_split ? need : bwd_split;
}

struct GramMse {  // the fused tail of maua_gram_fwd_mse_ledger (null target: plain maua_gram_fwd)
    const float* target;
    float* dmat;
    float loss_scale, grad_scale;
    double* rec;
};

// Row means (covariance form), the partial tiles of F F^T as split-K slabs in `workspace`, and - with many slabs - their first-level fold.
// *kstride_out = distance between the slabs the finishing pass has to add.
static int gram_partial_impl(const float* f, float* row_mean_out, int c, int64_t hw, int center, void* workspace, size_t workspace_bytes,
                             maua_stream_t stream, int* npairs_out, int* ksplit_out, int* kstride_out) {
    MAUA_REQUIRE(f && workspace && c > 0 && hw > 0 && c <= (1 << 16) && hw < (1ll << 30), MAUA_E_INVAL, "gram_fwd: bad args");
    MAUA_REQUIRE(!center || row_mean_out, MAUA_E_INVAL, "gram_fwd: center needs row_mean_out");
    MAUA_REQUIRE(workspace_bytes >= maua_gram_workspace_bytes(c, hw), MAUA_E_WORKSPACE, "gram_fwd: workspace %zu < %zu",
                 workspace_bytes, maua_gram_workspace_bytes(c, hw));
    int npairs, ksplit;
    int64_t chunk;
    gram_plan(c, hw, &npairs, &ksplit, &chunk);
    hipStream_t s = (hipStream_t)stream;
    if (center) {
        // the partial sums live at the start of the workspace, which the Gram slabs overwrite afterwards (same stream)
        hipLaunchKernelGGL(row_mean_partial_kernel, dim3(c, rm_split(hw)), dim3(256), 0, s, f, (double*)workspace, hw);
        int rc = check_launch("row_mean_partial_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(row_mean_finish_kernel, dim3((c + 255) / 256), dim3(256), 0, s, (const double*)workspace, row_mean_out,
                           c, hw);
        rc = check_launch("row_mean_finish_kernel");
        if (rc) return rc;
    }
    const bool use_x3 = tuning("gram_x3", 1) != 0;  // 0: the fp32-MFMA kernel (A/B comparisons)
    if (use_x3 && gram_tile128(c, hw)) {
        int wgs;
        gram_plan(c, hw, &npairs, &ksplit, &chunk, &wgs);
        static unsigned long long attr128 = 0;
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(gram_x3_partial128_kernel), GX128_LDS, &attr128);
        hipLaunchKernelGGL(gram_x3_partial128_kernel, dim3(gram128_grid(wgs, ksplit)), dim3(512), GX128_LDS, s, f, center ? row_mean_out : nullptr,
                           (float*)workspace, c, hw, ksplit, chunk, wgs);
    } else if (use_x3) {
        const int nplanes = c <= GT ? 2 : 4;
        const size_t lds = 2 * ((size_t)nplanes * GXPLANE + 64);
        static unsigned long long attr_set = 0;  // more than 64 KB of dynamic LDS needs the opt-in
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(gram_x3_partial_kernel), 2 * GXBUF, &attr_set);
        hipLaunchKernelGGL(gram_x3_partial_kernel, dim3(npairs, ksplit), dim3(256), lds, s, f, center ? row_mean_out : nullptr,
                           (float*)workspace, c, hw, ksplit, chunk, nplanes);
    }
    else
        hipLaunchKernelGGL(gram_partial_kernel, dim3(npairs, ksplit), dim3(256), 0, s, f, center ? row_mean_out : nullptr,
                           (float*)workspace, c, hw, ksplit, chunk);
    int rc = check_launch("gram_partial_kernel");
    if (rc) return rc;
    int kstride = 1;
    if (ksplit > 2 * GF_FOLD) {
        hipLaunchKernelGGL(gram_fold_kernel, dim3(npairs, GT * GT / 256, (ksplit + GF_FOLD - 1) / GF_FOLD), dim3(256), 0, s,
                           (float*)workspace, ksplit);
        rc = check_launch("gram_fold_kernel");
        if (rc) return rc;
        kstride = GF_FOLD;
    }
    *npairs_out = npairs;
    *ksplit_out = ksplit;
    *kstride_out = kstride;
    return MAUA_OK;
}

static int gram_fwd_impl(const float* f, float* gram, float* row_mean_out, int c, int64_t hw, float scale, int center,
                         void* workspace, size_t workspace_bytes, maua_stream_t stream, const GramMse& mse) {
    MAUA_REQUIRE(gram, MAUA_E_INVAL, "gram_fwd: bad args");
    int npairs, ksplit, kstride;
    int rc = gram_partial_impl(f, row_mean_out, c, hw, center, workspace, workspace_bytes, stream, &npairs, &ksplit, &kstride);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (mse.target) {
        hipLaunchKernelGGL(gram_finish_mse_kernel, dim3(npairs, GT * GT / 256), dim3(256), 0, s, (const float*)workspace, gram, c,
                           ksplit, kstride, scale, mse.target, mse.dmat, mse.grad_scale, mse.loss_scale, mse.rec);
        return check_launch("gram_finish_mse_kernel");
    }
    hipLaunchKernelGGL(gram_finish_kernel, dim3(npairs, GT * GT / 256), dim3(256), 0, s, (const float*)workspace, gram, c, ksplit,
                       kstride, scale);
    return check_launch("gram_finish_kernel");
}

int maua_gram_fwd(const float* f, float* gram, float* row_mean_out, int c, int64_t hw, float scale, int center,
                  void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    return gram_fwd_impl(f, gram, row_mean_out, c, hw, scale, center, workspace, workspace_bytes, stream, GramMse{});
}

int maua_gram_mse_ledger_supported(int c) {
    if (c <= 0 || c > (1 << 16)) return 0;
    const int64_t nt = (c + GT - 1) / GT;
    return nt * (nt + 1) / 2 * (GT * GT / 256) <= LEDGER_MAX ? 1 : 0;
}

int maua_gram_fwd_mse_ledger(const float* f, float* gram, float* row_mean_out, int c, int64_t hw, float scale, int center,
                             const float* target, float* dmat, float loss_scale, float grad_scale, double* ledger, int slot,
                             void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(target && dmat && ledger && slot >= 0 && slot < (1 << 28), MAUA_E_INVAL, "gram_fwd_mse_ledger: bad args");
    MAUA_REQUIRE(maua_gram_mse_ledger_supported(c), MAUA_E_UNSUPPORTED, "gram_fwd_mse_ledger: %d channels need more than %d partial sums",
                 c, LEDGER_MAX);
    GramMse m{target, dmat, loss_scale, grad_scale, ledger + (int64_t)slot * LEDGER_STRIDE};
    return gram_fwd_impl(f, gram, row_mean_out, c, hw, scale, center, workspace, workspace_bytes, stream, m);
}

int maua_gram_partial(const float* f, float* row_mean_out, int c, int64_t hw, int center, void* workspace, size_t workspace_bytes,
                      maua_stream_t stream) {
    int npairs, ksplit, kstride;
    return gram_partial_impl(f, row_mean_out, c, hw, center, workspace, workspace_bytes, stream, &npairs, &ksplit, &kstride);
}

int maua_gram_row_means(const float* f, float* row_mean_out, int c, int64_t hw, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(f && row_mean_out && workspace && c > 0 && hw > 0 && c <= (1 << 16) && hw < (1ll << 30), MAUA_E_INVAL, "gram_row_means: bad args");
    MAUA_REQUIRE(workspace_bytes >= (size_t)c * RM_SPLIT * sizeof(double), MAUA_E_WORKSPACE, "gram_row_means: workspace %zu < %zu", workspace_bytes,
                 (size_t)c * RM_SPLIT * sizeof(double));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_mean_partial_kernel, dim3(c, rm_split(hw)), dim3(256), 0, s, f, (double*)workspace, hw);
    int rc = check_launch("row_mean_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(row_mean_finish_kernel, dim3((c + 255) / 256), dim3(256), 0, s, (const double*)workspace, row_mean_out, c, hw);
    return check_launch("row_mean_finish_kernel");
}

int maua_gram_partial_batch(int count, const float* const* fs, float* const* means, const int* cs, const int64_t* hws,
                            void* const* workspaces, const size_t* workspace_bytes, const int* slab_counts, maua_stream_t stream) {
    MAUA_REQUIRE(count > 0 && count <= GB_MAX && fs && cs && hws && workspaces && workspace_bytes, MAUA_E_INVAL,
                 "gram_partial_batch: bad args (at most %d layers per call)", GB_MAX);
    for (int i = 0; i < count; ++i) {
        const int given = slab_counts ? slab_counts[i] : 0;  // > 0: the layer's slabs are in its workspace already (maua_conv3x3_image_gram)
        MAUA_REQUIRE((fs[i] || given > 0) && workspaces[i] && cs[i] > 0 && cs[i] <= (1 << 16) && hws[i] > 0 && hws[i] < (1ll << 30) && given >= 0 &&
                         given < (1 << 20),
                     MAUA_E_INVAL, "gram_partial_batch: bad args for layer %d", i);
        MAUA_REQUIRE(given == 0 || cs[i] <= GT, MAUA_E_UNSUPPORTED, "gram_partial_batch: ready slabs are for single-tile layers (layer %d)", i);
        const size_t need = given > 0 ? (size_t)given * GT * GT * sizeof(float) : maua_gram_workspace_bytes(cs[i], hws[i]);
        MAUA_REQUIRE(workspace_bytes[i] >= need, MAUA_E_WORKSPACE, "gram_partial_batch: workspace %zu < %zu (layer %d)", workspace_bytes[i], need, i);
    }
    const bool use_x3 = tuning("gram_x3", 1) != 0;
    if (!use_x3) {  // (the fp32-MFMA comparison kernel has no batched form)
        for (int i = 0; i < count; ++i) {
            MAUA_REQUIRE(!(slab_counts && slab_counts[i] > 0), MAUA_E_UNSUPPORTED, "gram_partial_batch: ready slabs need the fp16x3 kernels' fold");
            int rc = maua_gram_partial(fs[i], means ? means[i] : nullptr, cs[i], hws[i], means && means[i] ? 1 : 0, workspaces[i], workspace_bytes[i], stream);
            if (rc) return rc;
        }
        return MAUA_OK;
    }
    static unsigned long long attr_b = 0, attr_b128 = 0;
    (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(gram_x3_partial_batch_kernel), 2 * GXBUF, &attr_b);
    (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(gram_x3_partial128_batch_kernel), GX128_LDS, &attr_b128);
    hipStream_t s = (hipStream_t)stream;
    if (means) {  // covariance form: the row means of those layers first (two launches for all of them; their fp64 partial sums use the
                  // start of each layer's workspace, which its slabs overwrite afterwards)
        RowMeanBatch rm{};
        int nrm = 0, maxc = 0;
        for (int i = 0; i < count; ++i) {
            if (!means[i]) continue;
            MAUA_REQUIRE(fs[i] && !(slab_counts && slab_counts[i] > 0), MAUA_E_INVAL, "gram_partial_batch: row means need the layer's feature map (layer %d)", i);
            rm.f[nrm] = fs[i];
            rm.partial[nrm] = (double*)workspaces[i];
            rm.mean[nrm] = const_cast<float*>(means[i]);
            rm.HW[nrm] = hws[i];
            rm.C[nrm] = cs[i];
            rm.first[nrm + 1] = rm.first[nrm] + cs[i] * rm_split(hws[i]);
            maxc = cs[i] > maxc ? cs[i] : maxc;
            ++nrm;
        }
        if (nrm) {
            hipLaunchKernelGGL(row_mean_partial_batch_kernel, dim3(rm.first[nrm]), dim3(256), 0, s, rm);
            int rc = check_launch("row_mean_partial_batch_kernel");
            if (rc) return rc;
            hipLaunchKernelGGL(row_mean_finish_batch_kernel, dim3((maxc + 255) / 256, nrm), dim3(256), 0, s, rm);
            rc = check_launch("row_mean_finish_batch_kernel");
            if (rc) return rc;
        }
    }
    GramFoldBatch fold{};
    int nfold = 0, fold_pairs = 0;
    // three launches at most: the layers of one tile (C <= 64: two planes of LDS per buffer, four workgroups per CU), the other layers of
    // 64 x 64 blocks, and the layers of 128 x 128 blocks (one workgroup per CU)
    for (int group = 0; group < 3; ++group) {
        GramPartialBatch b{};
        int nb = 0;
        for (int i = 0; i < count; ++i) {
            const bool given = slab_counts && slab_counts[i] > 0;
            const int g = !given && gram_tile128(cs[i], hws[i]) ? 2 : cs[i] > GT ? 1 : 0;
            if (g != group) continue;
            int npairs, ksplit, wgs;
            int64_t chunk;
            gram_plan(cs[i], hws[i], &npairs, &ksplit, &chunk, &wgs);
            if (given) {  // nothing to multiply: only the first-level fold of the slabs that are there
                ksplit = slab_counts[i];
                if (ksplit > 2 * GF_FOLD) {
                    fold.partial[nfold] = (float*)workspaces[i];
                    fold.ksplit[nfold] = ksplit;
                    fold.npairs[nfold] = 1;
                    const int groups = (ksplit + GF_FOLD - 1) / GF_FOLD;
                    fold.max_groups = groups > fold.max_groups ? groups : fold.max_groups;
                    fold_pairs = 1 > fold_pairs ? 1 : fold_pairs;
                    ++nfold;
                }
                continue;
            }
            b.f[nb] = fs[i];
            b.mean[nb] = means ? means[i] : nullptr;
            b.partial[nb] = (float*)workspaces[i];
            b.HW[nb] = hws[i];
            b.chunk[nb] = chunk;
            b.C[nb] = cs[i];
            b.ksplit[nb] = ksplit;
            b.npairs[nb] = wgs;
            b.nplanes[nb] = group == 0 ? 2 : 4;
            b.first[nb + 1] = b.first[nb] + (group == 2 ? gram128_grid(wgs, ksplit) : wgs * ksplit);
            ++nb;
            if (ksplit > 2 * GF_FOLD) {
                fold.partial[nfold] = (float*)workspaces[i];
                fold.ksplit[nfold] = ksplit;
                fold.npairs[nfold] = npairs;
                const int groups = (ksplit + GF_FOLD - 1) / GF_FOLD;
                fold.max_groups = groups > fold.max_groups ? groups : fold.max_groups;
                fold_pairs = npairs > fold_pairs ? npairs : fold_pairs;
                ++nfold;
            }
        }
        if (!nb) continue;
        b.count = nb;
        if (group == 2) {
            hipLaunchKernelGGL(gram_x3_partial128_batch_kernel, dim3(b.first[nb]), dim3(512), GX128_LDS, s, b);
        } else {
            const size_t lds = 2 * ((size_t)(group == 1 ? 4 : 2) * GXPLANE + 64);
            hipLaunchKernelGGL(gram_x3_partial_batch_kernel, dim3(b.first[nb]), dim3(256), lds, s, b);
        }
        int rc = check_launch(group == 2 ? "gram_x3_partial128_batch_kernel" : "gram_x3_partial_batch_kernel");
        if (rc) return rc;
    }
    if (nfold) {
        hipLaunchKernelGGL(gram_fold_batch_kernel, dim3(fold_pairs, GT * GT / 256, nfold * fold.max_groups), dim3(256), 0, s, fold);
        return check_launch("gram_fold_batch_kernel");
    }
    return MAUA_OK;
}

int maua_gram_finish_mse_batch(int count, const void* const* workspaces, float* const* grams, const float* const* targets,
                               float* const* dmats, const int* cs, const int64_t* hws, const float* scales, const float* loss_scales,
                               const float* grad_scales, double* const* ledgers, const int* slots, const int* slab_counts,
                               maua_stream_t stream) {
    MAUA_REQUIRE(count > 0 && count <= GB_MAX && workspaces && grams && targets && dmats && cs && hws && scales && loss_scales &&
                     grad_scales && ledgers && slots,
                 MAUA_E_INVAL, "gram_finish_mse_batch: bad args (at most %d layers per call)", GB_MAX);
    GramFinishBatch b{};
    int max_pairs = 0;
    for (int i = 0; i < count; ++i) {
        MAUA_REQUIRE(workspaces[i] && grams[i] && targets[i] && dmats[i] && ledgers[i] && slots[i] >= 0 && slots[i] < (1 << 28) &&
                         cs[i] > 0 && cs[i] <= (1 << 16) && hws[i] > 0 && hws[i] < (1ll << 30),
                     MAUA_E_INVAL, "gram_finish_mse_batch: bad args for layer %d", i);
        MAUA_REQUIRE(maua_gram_mse_ledger_supported(cs[i]), MAUA_E_UNSUPPORTED, "gram_finish_mse_batch: %d channels need more than %d partial sums",
                     cs[i], LEDGER_MAX);
        int npairs, ksplit;
        int64_t chunk;
        gram_plan(cs[i], hws[i], &npairs, &ksplit, &chunk);
        if (slab_counts && slab_counts[i] > 0) {  // (slabs of maua_conv3x3_image_gram: their number is the launch's, not the plan's)
            MAUA_REQUIRE(cs[i] <= GT && slab_counts[i] < (1 << 20), MAUA_E_UNSUPPORTED, "gram_finish_mse_batch: ready slabs are for single-tile layers");
            ksplit = slab_counts[i];
        }
        b.partial[i] = (const float*)workspaces[i];
        b.gram[i] = grams[i];
        b.target[i] = targets[i];
        b.dmat[i] = dmats[i];
        b.rec[i] = ledgers[i] + (int64_t)slots[i] * LEDGER_STRIDE;
        b.C[i] = cs[i];
        b.ksplit[i] = ksplit;
        b.kstride[i] = ksplit > 2 * GF_FOLD ? GF_FOLD : 1;
        b.npairs[i] = npairs;
        b.scale[i] = scales[i];
        b.grad_scale[i] = grad_scales[i];
        b.loss_scale[i] = loss_scales[i];
        max_pairs = npairs > max_pairs ? npairs : max_pairs;
    }
    hipLaunchKernelGGL(gram_finish_mse_batch_kernel, dim3(max_pairs, GT * GT / 256, count), dim3(256), 0, (hipStream_t)stream, b);
    return check_launch("gram_finish_mse_batch_kernel");
}

int maua_gram_bwd(const float* d_sym, const float* f, const float* row_mean, const float* relu_mask, float* gf, int c,
                  int64_t hw, int accumulate, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(d_sym && f && gf && c > 0 && hw > 0 && c <= (1 << 16) && hw < (1ll << 30), MAUA_E_INVAL, "gram_bwd: bad args");
    MAUA_REQUIRE(hw < (1ll << 31), MAUA_E_UNSUPPORTED, "gram_bwd: plane too large");
    const bool use_x3 = tuning("gram_bwd_x3", 1) != 0;
    if (use_x3 && hw < (1ll << 30)) {
        // fp16x3 product (conv1x1_x3.hip): D is symmetric, so its rows serve as [cout][cin]; centring folded into the staging
        ConvArgs a{};
        a.x = f;
        a.w = d_sym;
        a.omask = relu_mask;
        a.y = gf;
        a.Cin = c;
        a.Cout = c;
        a.H = 1;
        a.W = (int)hw;
        a.accumulate = accumulate;
        const size_t need = conv1x1_x3_workspace(1, c, hw, c);
        a.ws = (workspace && need && workspace_bytes >= need) ? (float*)workspace : nullptr;
        return conv1x1_x3_launch(a, row_mean, 1, (hipStream_t)stream);
    }
    if (hw % 4 == 0 && c % 4 == 0 && (uintptr_t)f % 16 == 0 && (uintptr_t)d_sym % 16 == 0) {
        constexpr size_t lds = 2ull * (GB_KC * GB_PX + GB_KC * GB_CO) * sizeof(float);
        dim3 grid((unsigned)((hw + GB_PX - 1) / GB_PX), (unsigned)((c + GB_CO - 1) / GB_CO));
        auto launch = [&](auto kernel) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kernel, grid, dim3(256), lds, (hipStream_t)stream, d_sym, f, row_mean, relu_mask, gf, c, hw,
                               accumulate);
        };
        if (accumulate && relu_mask) launch(gram_bwd_kernel<true, true>);
        else if (accumulate) launch(gram_bwd_kernel<true, false>);
        else if (relu_mask) launch(gram_bwd_kernel<false, true>);
        else launch(gram_bwd_kernel<false, false>);
        return check_launch("gram_bwd_kernel");
    }
    // general shapes: the 1x1 path of the convolution kernel, centring as a per-row bias
    float* bias = nullptr;
    if (row_mean) {
        MAUA_REQUIRE(workspace && workspace_bytes >= (size_t)c * sizeof(float), MAUA_E_WORKSPACE,
                     "gram_bwd: workspace too small for the centering bias");
        bias = (float*)workspace;
        hipLaunchKernelGGL(center_bias_kernel, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_sym, row_mean,
                           bias, c);
        int rc = check_launch("center_bias_kernel");
        if (rc) return rc;
    }
    ConvArgs a{};
    a.x = f;
    a.mask = nullptr;
    a.w = d_sym;  // symmetric: [k][c] layout == [c][k]
    a.bias = bias;
    a.omask = relu_mask;
    a.y = gf;
    a.Cin = c;
    a.Cout = c;
    a.H = 1;
    a.W = (int)hw;
    a.OH = 1;
    a.OW = (int)hw;
    a.pad = 0;
    a.relu = 0;
    a.accumulate = accumulate;
    return conv_mfma_dispatch(a, 1, 1, (hipStream_t)stream);
}

}  // extern "C"
