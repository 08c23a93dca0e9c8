// 3x3 stride-1 convolution with fp32-level accuracy on the fp16 matrix cores ("fp16x3").
//
// An fp32 value scaled by a power of two splits into two fp16 parts, x s = xh + xl, that together carry 22 of its 24
// significant bits (round-to-nearest both times); a product of two such numbers is
//        xh*wh + (xh*wl + xl*wh)                      (+ xl*wl, below 2^-22: dropped)
// with every term exact in fp32.  Measured against fp64 on layer-shaped data the representation error of the result is
// 7e-8 relative - below the 1.6e-7 of an fp32 FMA chain - so three v_mfma_f32_32x32x16_f16 do the work of the six bf16
// MFMAs of conv_x6.hip (on a power-bound kernel: half the matrix energy).
//
// fp16 has 5 exponent bits, so the scale matters.  Filters get one power-of-two scale per layer (|w s_w| < 64, chosen by
// the host when the bank is packed).  Activations / gradients are scaled PER WORKGROUP AND PER 8-CHANNEL CHUNK: the
// workgroup takes the maximum of the chunk it is about to stage (wave reduction + four LDS words, behind a barrier the
// pipeline has anyway), scales it into [2^11, 2^12), and un-scales the chunk's partial sums when it folds them into the
// fp32 master accumulator (the fold happens every chunk already, conv_x6.hip).  Elements within 2^15 of their chunk's
// maximum keep all 22 bits; smaller ones degrade gracefully (their low part becomes subnormal), i.e. the error stays
// relative to the largest values of the tile - the same behaviour as fp32 accumulation.  Nothing can overflow.
//
// Everything else follows conv_x6.hip: workgroup = 4 waves = 64 output channels x (4 rows x 32 px), K chunks of 8 input
// channels, one k-step = two taps (the ninth alone at K = 8), one LDS chunk (patch [part][pos][8ch] 6.5 KB + filters
// [tap][part][co][8ch] 18 KB), filter halves streamed by LDS-DMA behind the k-steps, counted vmcnt, XCD-aware tile
// order, deterministic split-K.  The fp32 master accumulator is always present here (it is where the per-chunk scales
// meet), so one variant serves every channel count (conv1_1's three channels are a single, partly empty chunk).

// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int X3_COT = 64;
constexpr int X3_PH = 4, X3_PR = 6, X3_PC = 34;
constexpr int X3_NPOS = X3_PR * X3_PC;                    // 204
constexpr int X3_NPOS_PAD = 208;
constexpr int X3_PATCH_BYTES = 2 * X3_NPOS_PAD * 16;      // 6656
constexpr int X3_W_BYTES = 9 * 2 * X3_COT * 16;           // 18432 = 18 planes of 1 KiB: [tap][part][co][16 B]

// two floats -> packed fp16 pair, round to nearest even (bits 15:0 = a)
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// bank[dir][chunk][cotile][tap][part][co][ch] (fp16, pre-scaled by w_scale): fwd: co = output channel, ch = input
// channel, tap = ky*3+kx; bwd-data: roles swapped and taps flipped.  Zero padding for channels beyond the tensor.
__global__ void pack_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int cout, int cin, int backward,
                               float w_scale) {
    const int CO = backward ? cin : cout;
    const int CI = backward ? cout : cin;
    const int nchunk = (CI + 7) / 8, ntile = (CO + X3_COT - 1) / X3_COT;
    const int64_t total = (int64_t)nchunk * ntile * 9 * X3_COT * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int ch = (int)(r % 8);
        r /= 8;
        const int co = (int)(r % X3_COT);
        r /= X3_COT;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * X3_COT + co, i = chunk * 8 + ch;
        float v = 0.f;
        if (o < CO && i < CI) {
            if (!backward) v = w[((int64_t)o * cin + i) * 9 + tap];
            else v = w[((int64_t)i * cin + o) * 9 + (8 - tap)];
        }
        v *= w_scale;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        const int64_t base = (((((int64_t)chunk * ntile + tile) * 9 + tap) * 2) * X3_COT + co) * 8 + ch;
        bank[base] = __builtin_bit_cast(unsigned short, h);
        bank[base + X3_COT * 8] = __builtin_bit_cast(unsigned short, l);
    }
}

template <bool ACC, bool OM, bool C8>
__global__ void __launch_bounds__(256, 4) conv_x3_kernel(ConvArgs p, float w_inv_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[X3_PATCH_BYTES + X3_W_BYTES + 16];
    unsigned char* Pl = smem;                    // [part][pos][16 B]
    unsigned char* Wl = smem + X3_PATCH_BYTES;   // [tap][part][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + X3_PATCH_BYTES + X3_W_BYTES);  // per-wave maxima of the chunk being staged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    const int cotile = blockIdx.y;
    const int co0 = cotile * X3_COT;
    const int ntile = gridDim.y;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    // XCD-aware tile order: XCD k (workgroups are dealt round-robin by linear id) owns the k-th contiguous band of tiles
    const int tiles_total = p.tiles_x * ((p.OH + X3_PH - 1) / X3_PH);
    const int per_xcd = (tiles_total + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, tiles_total)) return;  // whole workgroup leaves
    const int x0 = (tile % p.tiles_x) * 32, y0 = (tile / p.tiles_x) * X3_PH;

    unsigned p_byte = 0;
    bool pos_ok = false;
    {
        const int r = tid / X3_PC, col = tid - r * X3_PC;
        const int iy = y0 + r - p.pad, ix = x0 + col - p.pad;
        pos_ok = tid < X3_NPOS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        p_byte = pos_ok ? (unsigned)(iy * p.W + ix) * 4u : 0u;
    }
    float rp[8];
    int rp_c0 = 0;
    auto load_patch = [&](int c0) {
        asm volatile("" : "+s"(c0));
        rp_c0 = c0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chn = C8 ? c0 + c : min(c0 + c, p.Cin - 1);
            const char* plane = reinterpret_cast<const char*>(xin + (int64_t)chn * in_plane);
            rp[c] = *reinterpret_cast<const float*>(plane + p_byte);
        }
    };
    // this wave's maximum of the staged chunk -> LDS (read by everybody after the next barrier)
    auto publish_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool ok = pos_ok && (C8 || rp_c0 + c < p.Cin);
            m = fmaxf(m, ok ? fabsf(rp[c]) : 0.f);
        }
        m = wave_max_nonneg(m);
        if (lane == 0) Ml[wave] = m;
    };
    // scale of the staged chunk: max in [2^11, 2^12) after scaling.  Returns the INVERSE scale, sets `sx`.
    float sx = 1.f;
    auto chunk_scale = [&]() {
        const float m = fmaxf(fmaxf(Ml[0], Ml[1]), fmaxf(Ml[2], Ml[3]));
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;  // floor(log2 m) for normal m
        e = m > 0.f ? max(e, -100) : 11;
        sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
        return __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23);
    };
    auto store_patch = [&]() {
        if (tid < X3_NPOS_PAD) {
            u32x4 vh, vl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok0 = pos_ok && (C8 || rp_c0 + 2 * q < p.Cin), ok1 = pos_ok && (C8 || rp_c0 + 2 * q + 1 < p.Cin);
                const float v0 = ok0 ? rp[2 * q] * sx : 0.f, v1 = ok1 ? rp[2 * q + 1] * sx : 0.f;
                const unsigned H = cvt_pk_f16(v0, v1);
                vh[q] = H;
                vl[q] = cvt_pk_f16(v0 - f16_lo(H), v1 - f16_hi(H));
            }
            *reinterpret_cast<u32x4*>(Pl + (0 * X3_NPOS_PAD + tid) * 16) = vh;
            *reinterpret_cast<u32x4*>(Pl + (1 * X3_NPOS_PAD + tid) * 16) = vl;
        }
    };

    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    // Filter slice of a chunk = 18 planes of 1 KiB in LDS order.  Half A = taps 0-3 (planes 0-7, read by k-steps 0-1),
    // half B = taps 4-8 (planes 8-17, read by k-steps 2-4).  Every wave issues exactly 2 (A) or 3 (B) LDS-DMA instructions
    // so that the counted vmcnt waits are the same for all waves (waves 2 and 3 repeat plane 17).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane16 = lane * 16;
    auto dma_half = [&](int ch, bool second) {
        const unsigned char* src = bank + ((int64_t)ch * ntile + cotile) * X3_W_BYTES;
        const int first = second ? 8 : 0, count = second ? 3 : 2, last = second ? 17 : 7;
        for (int i = 0; i < count; ++i) {
            const int q = min(first + wv + 4 * i, last);
            const unsigned char* g = src + q * 1024;
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(Wl + q * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(g)
                         : "memory");
        }
    };

    // fragment addresses: tap of this lane half for k-step s < 4 is 2s + half; the ninth tap runs alone at K = 8
    int b_byte[5], a_byte[5];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tap = 2 * s + half;
        const int ky = tap / 3, kx = tap - 3 * ky;
        b_byte[s] = ((wave + ky) * X3_PC + j + kx) * 16;
        a_byte[s] = tap * 2048 + j * 16;
    }
    b_byte[4] = ((wave + 2) * X3_PC + j + 2) * 16 + half * 8;
    a_byte[4] = 8 * 2048 + j * 16 + half * 8;

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
                acc[t][r] = 0.f;
            }
    }

    auto kstep = [&](int s) {
        f16x8 b[2], a[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) b[part] = *reinterpret_cast<const f16x8*>(Pl + part * X3_NPOS_PAD * 16 + b_byte[s]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                a[t][part] = *reinterpret_cast<const f16x8*>(Wl + a_byte[s] + part * 1024 + t * 512);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[0], acc[t], 0, 0, 0);  // smallest terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };
    auto kstep_tap9 = [&]() {
        f16x4 b[2], a[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) b[part] = *reinterpret_cast<const f16x4*>(Pl + part * X3_NPOS_PAD * 16 + b_byte[4]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                a[t][part] = *reinterpret_cast<const f16x4*>(Wl + a_byte[4] + part * 1024 + t * 512);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };

    // Pipeline as in conv_x6.hip, with the chunk's scale in between:
    //   chunk c: [patch(c+1) loads] ks0 ks1 | X1 | DMA A(c+1) | ks2 ks3 ks4, fold acc (x inv scale of c), publish max(c+1) |
    //            X2 | scale(c+1), split + write patch(c+1), DMA B(c+1), wait A(c+1) | X3 | next chunk
    // vmcnt: before X1 the queue is [B(c) x3, patch loads x8] -> vmcnt(8); before X3 [A(c+1) x2, B(c+1) x3] -> vmcnt(3).
    const int nchunks_all = (p.Cin + 7) / 8;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);
    dma_half(ch_begin, false);
    dma_half(ch_begin, true);
    load_patch(ch_begin * 8);
    publish_max();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float inv_cur = chunk_scale() * w_inv_scale;  // un-scaling factor of the chunk whose products are accumulating
    store_patch();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int tg_slot = (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4) & 3);  // hwreg(HW_REG_HW_ID, 16, 4)
    for (int ch = ch_begin; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        switch ((tg_slot + ch) & 3) {  // rotate the issue priority among the workgroups of a CU (conv_x6.hip)
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        if (more) load_patch((ch + 1) * 8);
        kstep(0);
        kstep(1);
        if (more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X1: B(ch) landed, A(ch)'s readers done
        if (more) dma_half(ch + 1, false);
        kstep(2);
        kstep(3);
        kstep_tap9();
        if (more) publish_max();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                master[t][r] = fmaf(acc[t][r], inv_cur, master[t][r]);  // fold + un-scale (power of two: exact)
                acc[t][r] = 0.f;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X2: every wave is done reading the patch and half B; the maxima are visible
        if (more) {
            inv_cur = chunk_scale() * w_inv_scale;
            store_patch();
            dma_half(ch + 1, true);
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");  // A(ch+1) landed, patch writes done
            __builtin_amdgcn_s_barrier();  // X3
        }
    }

    // epilogue (conv_x6.hip): lane holds pixel column j of row y0+wave; register r is output channel (r&3)+8*(r>>2)+4*half
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
    const int oy = y0 + wave, ox = x0 + j;
    const bool pvalid = oy < p.OH && ox < p.OW;
    const int64_t opix = (int64_t)oy * p.OW + ox;
    if (p.ksplit > 1) {  // split-K: un-scaled partial sums, finished by conv_splitk_finish_kernel in split order
        float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * out_plane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pvalid && co < p.Cout) wsp[(int64_t)co * out_plane + opix] = master[t][r];
            }
        return;
    }
    if (pvalid) {
        const bool full = co0 + X3_COT <= p.Cout;
        const int64_t lane_off = (int64_t)(co0 + 4 * half) * out_plane + opix;
        float* __restrict__ yl = yout + lane_off;
        const float* __restrict__ oml = OM ? om + lane_off : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float prev[16], msk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                const int64_t o = (full || co0 + cr + 4 * half < p.Cout) ? (int64_t)cr * out_plane : 0;
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = yl[o];
                if constexpr (OM) msk[r] = oml[o];
            }
            float outv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = master[t][r];
                v += prev[r];
                if (p.relu) v = v > 0.f ? v : 0.f;
                outv[r] = msk[r] > 0.f ? v : 0.f;
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yl[(int64_t)(t * 32 + (r & 3) + 8 * (r >> 2)) * out_plane] = outv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                    if (co0 + cr + 4 * half < p.Cout) yl[(int64_t)cr * out_plane] = outv[r];
                }
            }
        }
    }
}

// same cost model as conv_x6.hip with this kernel's ~2.7 us per chunk
static int x3_choose_split(const ConvArgs& a, int n) {
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + X3_PH - 1) / X3_PH) * ((a.Cout + X3_COT - 1) / X3_COT) * split_batch_hint();  // planned frames (see conv_x3w.hip)
    const int nchunks = (a.Cin + 7) / 8;
    if (wgs >= 4096 || nchunks < 8) return 1;
    const double out_mb = (double)split_batch_hint() * a.Cout * a.OH * a.OW * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 4; ++ks) {
        const double rounds = (double)((wgs * ks + 1023) / 1024);
        double cost = rounds * ((double)((nchunks + ks - 1) / ks) + 2.0) * 2.7;
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

int conv_x3_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
    p.tiles_x = (a.OW + 31) / 32;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + X3_PH - 1) / X3_PH);
    const int ks = a.ws ? x3_choose_split(a, n) : 1;
    p.ksplit = ks;
    const int64_t cot = (a.Cout + X3_COT - 1) / X3_COT, per_xcd = (tiles + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8), (unsigned)cot, (unsigned)(n * ks));
    const bool acc = ks == 1 && a.accumulate != 0, om = ks == 1 && a.omask != nullptr;
    const float w_inv = 1.f / w_scale;
#define MAUA_X3_LAUNCH(ACC_, OM_)                                                                                 \
    do {                                                                                                          \
        if (a.Cin % 8 == 0) hipLaunchKernelGGL((conv_x3_kernel<ACC_, OM_, true>), grid, dim3(256), 0, stream, p, w_inv);   \
        else hipLaunchKernelGGL((conv_x3_kernel<ACC_, OM_, false>), grid, dim3(256), 0, stream, p, w_inv);                 \
    } while (0)
    if (acc && om) MAUA_X3_LAUNCH(true, true);
    else if (acc) MAUA_X3_LAUNCH(true, false);
    else if (om) MAUA_X3_LAUNCH(false, true);
    else MAUA_X3_LAUNCH(false, false);
#undef MAUA_X3_LAUNCH
    int rc = check_launch("conv_x3_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_x3_bank_bytes(int cout_produced, int cin_consumed) {
    if (cout_produced <= 0 || cin_consumed <= 0 || cout_produced > (1 << 20) || cin_consumed > (1 << 20)) return 0;
    const size_t nchunk = (cin_consumed + 7) / 8, ntile = (cout_produced + X3_COT - 1) / X3_COT;
    return nchunk * ntile * X3_W_BYTES;
}

int maua_conv_pack_filters_x3(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                              maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20) && w_scale > 0.f, MAUA_E_INVAL,
                 "conv_pack_filters_x3: bad args");
    if (bank_fwd) {
        hipLaunchKernelGGL(pack_x3_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_fwd, cout,
                           cin, 0, w_scale);
        int rc = check_launch("pack_x3_kernel");
        if (rc) return rc;
    }
    if (bank_bwd) {
        hipLaunchKernelGGL(pack_x3_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_bwd, cout,
                           cin, 1, w_scale);
        return check_launch("pack_x3_kernel");
    }
    return MAUA_OK;
}

size_t maua_conv_x3_workspace_bytes(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const int ks = x3_choose_split(a, n);
    return ks > 1 ? (size_t)n * ks * cout * a.OH * a.OW * sizeof(float) : 0;
}

int maua_conv3x3_x3(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                    int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3: bad dims");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_x3: input smaller than the filter");
    MAUA_REQUIRE((int64_t)h * w < (1ll << 30), MAUA_E_UNSUPPORTED, "conv3x3_x3: plane too large");
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
    a.ws = (workspace && workspace_bytes >= maua_conv_x3_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    return conv_x3_launch(a, n, w_scale, (hipStream_t)stream);
}

}  // extern "C"
