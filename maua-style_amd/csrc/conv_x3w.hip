// 3x3 stride-1 convolution in fp16x3 arithmetic (conv_x3.hip), second structure: "wide" register tiles, 16-channel chunks.
//
// conv_x3.hip runs four workgroups per CU (four waves per SIMD, 128 registers each): every MFMA is fed by one ds_read_b128,
// every 8-channel chunk pays three workgroup barriers and a fold of the chunk's sums into the fp32 master accumulator, and
// the ninth tap runs as a half-empty K = 8 step (10 % of the matrix-pipe time).  The kernel sits at 57 % pipe occupancy and
// the chip holds ~1.8 GHz under it: what is left to win is energy and issue slots per useful product.
// This structure spends the register file differently:
//   * workgroup = 4 waves = 64 output channels x (8 rows x 32 px); a wave owns 64 channels x 2 rows x 32 px = FOUR 32x32
//     accumulators (+ four fp32 masters): 128 accumulator registers, two workgroups per CU (two waves per SIMD, 256
//     registers each).  One tap needs 4 filter fragments + 4 patch fragments for 12 MFMAs: 0.67 ds_read_b128 per MFMA
//     instead of 1;
//   * K chunk = 16 input channels: one MFMA k-step is ONE tap x 16 channels (lane half = channel octet), so all nine taps
//     are full K = 16 steps - no K = 8 step; the fold into the masters and the workgroup's scale happen once per 16
//     channels, with TWO barriers per chunk (three per 8 channels before);
//   * image borders and the tail items of the staging loop are out-of-range BUFFER loads (the hardware returns 0): no
//     per-value selects while staging;
//   * the two workgroups of a CU are independent, so while one stages (split + LDS writes + filter DMA) the other owns the
//     matrix pipe.
// Arithmetic is exactly conv_x3.hip's: x s = xh + xl (two fp16 parts, round to nearest, s = power of two per workgroup and
// chunk bringing the chunk's maximum into [2^11, 2^12)), filters pre-split and pre-scaled in the bank, products xl*wh, xh*wl,
// xh*wh on v_mfma_f32_32x32x16_f16, per-chunk un-scaled fold into fp32 masters.  The scale now covers 16 channels.
//
// LDS per workgroup: patch [part][octet][pos 10x34 (+4)][8 ch] = 22,016 B, filters [tap][part][octet][co 64][8 ch] = 36,864 B.
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>

#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int XW_COT = 64;
constexpr int XW_ROWS = 8, XW_PR = 10, XW_PC = 34;
constexpr int XW_NPOS = XW_PR * XW_PC;                          // 340
constexpr int XW_NPOS_PAD = 344;
constexpr int XW_PLANE = XW_NPOS_PAD * 16;                      // bytes of one [pos][8 ch] plane
constexpr int XW_PATCH_BYTES = 4 * XW_PLANE;                    // [part][octet]
constexpr int XW_W_BYTES = 9 * 2 * 2 * XW_COT * 16;             // 36 planes of 1 KiB: [tap][part][octet][co][16 B]
constexpr int XW_ITEMS = 2 * XW_NPOS;                           // (octet, pos) staging items per chunk: 680 = 2.66 per thread

__device__ __forceinline__ unsigned xw_cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float xw_f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float xw_f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// bank[dir][chunk16][cotile][tap][part][octet][co][ch] (fp16, pre-scaled): fwd: co = output channel, ch = input channel,
// tap = ky*3+kx; bwd-data: roles swapped and taps flipped.  Zero padding for channels beyond the tensor.
__global__ void pack_x3w_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int cout, int cin, int backward,
                                float w_scale) {
    const int CO = backward ? cin : cout;
    const int CI = backward ? cout : cin;
    const int nchunk = (CI + 15) / 16, ntile = (CO + XW_COT - 1) / XW_COT;
    const int64_t total = (int64_t)nchunk * ntile * 9 * 2 * XW_COT * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int ch = (int)(r % 8);
        r /= 8;
        const int co = (int)(r % XW_COT);
        r /= XW_COT;
        const int oct = (int)(r % 2);
        r /= 2;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * XW_COT + co, i = chunk * 16 + oct * 8 + ch;
        float v = 0.f;
        if (o < CO && i < CI) {
            if (!backward) v = w[((int64_t)o * cin + i) * 9 + tap];
            else v = w[((int64_t)i * cin + o) * 9 + (8 - tap)];
        }
        v *= w_scale;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        // [tap][part][octet][co][ch]
        const int64_t base = ((int64_t)chunk * ntile + tile) * (XW_W_BYTES / 2);
        bank[base + ((((int64_t)tap * 2 + 0) * 2 + oct) * XW_COT + co) * 8 + ch] = __builtin_bit_cast(unsigned short, h);
        bank[base + ((((int64_t)tap * 2 + 1) * 2 + oct) * XW_COT + co) * 8 + ch] = __builtin_bit_cast(unsigned short, l);
    }
}

// The symmetric C x C matrix D of a style loss (grad_scale (G - T)) as ONE-tap filter bank for the fused Gram-backward term of
// conv_x3w_kernel: bank[chunk16][cotile][part][octet][co][ch] = D[co][ch] scaled by a power of two that brings max |D| into
// [32, 64) like the 3x3 banks; inv[0] = 1 / scale.  Every workgroup finds the maximum itself (C <= 512: 1 MB out of L2), then
// packs its share: a thread takes the 8 consecutive values of one (chunk, tile, octet, co) and writes their two 16-byte halves.
__device__ __forceinline__ void pack_dmat_x3w_body(const float* __restrict__ d, int C, unsigned short* __restrict__ bank,
                                                   float* __restrict__ inv) {
    __shared__ float wmax[16];
    const int tid = threadIdx.x;
    const int64_t total4 = (int64_t)C * C / 4;  // (C % 16 == 0)
    float m = 0.f;
    // (all loads of the maximum in flight together - a 256 x 256 matrix is sixteen per thread: the launch was a chain of sixteen round trips)
    int64_t e = tid;
    for (; e + 7 * 1024 < total4; e += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(d)[e + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (; e < total4; e += 1024) {
        const float4 v = reinterpret_cast<const float4*>(d)[e];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max_nonneg(m);
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    m = 0.f;
    for (int i = 0; i < 16; ++i) m = fmaxf(m, wmax[i]);
    float scale = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
        scale = __builtin_bit_cast(float, (unsigned)(127 + 5 - max(e, -100)) << 23);
    }
    if (tid == 0 && blockIdx.x == 0) inv[0] = 1.f / scale;
    const int nchunk = C / 16, ntile = (C + XW_COT - 1) / XW_COT;
    const int64_t groups = (int64_t)nchunk * ntile * 2 * XW_COT;  // (chunk, tile, octet, co)
    for (int64_t e = (int64_t)blockIdx.x * 1024 + tid; e < groups; e += (int64_t)gridDim.x * 1024) {
        int64_t r = e;
        const int co = (int)(r % XW_COT);
        r /= XW_COT;
        const int oct = (int)(r % 2);
        r /= 2;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * XW_COT + co, i = chunk * 16 + oct * 8;
        u32x4 Hh = {0u, 0u, 0u, 0u}, Ll = {0u, 0u, 0u, 0u};
        if (o < C) {
            const float4 v0 = *reinterpret_cast<const float4*>(d + (int64_t)o * C + i);
            const float4 v1 = *reinterpret_cast<const float4*>(d + (int64_t)o * C + i + 4);
            const float v[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned H = xw_cvt_pk_f16(v[2 * q], v[2 * q + 1]);
                Hh[q] = H;
                Ll[q] = xw_cvt_pk_f16(v[2 * q] - xw_f16_lo(H), v[2 * q + 1] - xw_f16_hi(H));
            }
        }
        unsigned char* base = reinterpret_cast<unsigned char*>(bank) + ((int64_t)chunk * ntile + tile) * 4096 + co * 16;  // [part][octet][co][16 B]
        *reinterpret_cast<u32x4*>(base + (0 * 2 + oct) * 1024) = Hh;
        *reinterpret_cast<u32x4*>(base + (1 * 2 + oct) * 1024) = Ll;
    }
}
__global__ void __launch_bounds__(1024) pack_dmat_x3w_kernel(const float* __restrict__ d, int C, unsigned short* __restrict__ bank,
                                                             float* __restrict__ inv) {
    pack_dmat_x3w_body(d, C, bank, inv);
}
// ... of up to four style layers in one launch (grid y = layer; every layer's workgroups stride over its own groups)
struct DmatPackBatch {
    const float* d[4];
    unsigned short* bank[4];
    float* inv[4];
    int C[4];
};
__global__ void __launch_bounds__(1024) pack_dmat_x3w_batch_kernel(DmatPackBatch b) {
    pack_dmat_x3w_body(b.d[blockIdx.y], b.C[blockIdx.y], b.bank[blockIdx.y], b.inv[blockIdx.y]);
}

// Diagnostic build only (-DXW_STAMP, tools/x3w_clock.py): shader-clock stamps at the phase boundaries of every chunk go to a
// buffer of their own (passed in p.mask, which this kernel does not use otherwise); no output value depends on them.
#ifdef XW_STAMP
#define XW_MARK(k)                                                                                          \
    do {                                                                                                    \
        if (lane == 0 && p.mask) {                                                                          \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                     \
            const_cast<float*>(p.mask)[((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 64 + (ch - ch_begin)) * 8 + (k)] = \
                __builtin_bit_cast(float, (unsigned)t_);                                                    \
        }                                                                                                   \
    } while (0)
#else
#define XW_MARK(k) do {} while (0)
#endif

// UNPOOL: `x` is the POOLED map of a 2x2 / 2 max pool and `in_codes` its decision bytes; the input the convolution sees is the pool's
// backward pass over them (pool2x2_bwd_codes_kernel's arithmetic), rebuilt while staging - the full-size gradient never exists.
// The kernel's argument segment (ConvArgs is the first parameter): the in-launch finish re-reads its arguments from here behind the K loop
__device__ __forceinline__ const ConvArgs* xw_kernel_arguments() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}

template <bool ACC, bool OM, bool POOL = false, bool UNPOOL = false>
__global__ void __launch_bounds__(256, 2) conv_x3w_kernel(ConvArgs p, float w_inv_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[XW_PATCH_BYTES + XW_W_BYTES + 32];  // (+ four maxima, the in-launch finish's three words)
    unsigned char* Pl = smem;                    // [part][octet][pos][16 B]
    unsigned char* Wl = smem + XW_PATCH_BYTES;   // [tap][part][octet][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + XW_PATCH_BYTES + XW_W_BYTES);  // per-wave maxima of the chunk being staged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    // XCD-aware order (conv_x6.hip): XCD k owns the k-th contiguous band of pixel tiles.  Workgroups go to the XCDs in turn in dispatch order
    // (x fastest, gridDim.x % 8 == 0), so `slot` counts an XCD's workgroups in the order they start: with p.cot_inner the channel tiles of a
    // pixel tile take consecutive slots - they run side by side on one XCD and the patch leaves memory once, not once per channel tile.
    const int ntile = gridDim.y;
    const int tiles_total = p.tiles_x * ((p.OH + XW_ROWS - 1) / XW_ROWS);
    const int per_xcd = (tiles_total + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = (int)((blockIdx.y * gridDim.x + blockIdx.x) >> 3);
    const int cotile = p.cot_inner ? slot % ntile : (int)blockIdx.y;
    const int tile = xcd * per_xcd + (p.cot_inner ? slot / ntile : (int)(blockIdx.x >> 3));
    const int co0 = cotile * XW_COT;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const int st_w = UNPOOL ? p.W >> 1 : p.W;                              // row pitch and plane of the array the patch is staged from
    const int st_plane = UNPOOL ? (p.H >> 1) * (p.W >> 1) : in_plane;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * st_plane;
    const unsigned char* __restrict__ xcodes = UNPOOL ? p.in_codes + (int64_t)n * p.Cin * st_plane : nullptr;
    const unsigned code_mask = (unsigned)p.in_code_mask;
    if (tile >= min((xcd + 1) * per_xcd, tiles_total)) return;  // whole workgroup leaves
    const int x0 = (tile % p.tiles_x) * 32, y0 = (tile / p.tiles_x) * XW_ROWS;
    if (tid == 0) {  // for the in-launch finish of a split channel loop (behind the K loop; read there behind several barriers)
        reinterpret_cast<int*>(Ml)[5] = (n * ntile + cotile) * tiles_total + tile;
        reinterpret_cast<int*>(Ml)[6] = split;
    }

    // Staging items of this thread: item k = (octet, position) number tid + 256 k.  voff = byte offset of the item's first
    // channel from the chunk's first plane; out-of-image positions and items past the end get an offset beyond the buffer's
    // range, for which a buffer load returns 0 (no selects on the values).
    //
    // UNPOOL: one item per thread, and it is a POOLED element - (octet, pooled row, pooled column) number tid of the 2 x 6 x 18 pooled
    // elements whose 2x2 windows cover the 10 x 34 patch.  The thread loads its eight channels and their eight decision bytes (one
    // 8-byte load from the [octet][pooled pixel][8] layout), splits the eight values once, and writes each of the window's four
    // corners that lies inside the patch: channel c's halves where the byte names that corner, zero elsewhere - the patch in LDS is
    // bit for bit what staging pool2x2_bwd_codes_kernel's output would have left (a quarter of the loads and splits, no full-size
    // gradient in memory).
    unsigned voff[3], lds_w[3];
    unsigned vcode = 0x80000000u, inmask = 0;  // UNPOOL: offset of the item's decision bytes; bit q = corner q (2 dy + dx) is a patch position
    if constexpr (UNPOOL) {
        constexpr int PR = XW_ROWS / 2 + 2, PCW = 18;  // pooled rows / columns under the patch (any pad in 0 ... 2)
        const int oct = tid >= PR * PCW ? 1 : 0;
        const int rem = tid - oct * PR * PCW;
        const int pr = rem / PCW, pc = rem - pr * PCW;
        const int ppy = ((y0 - p.pad) >> 1) + pr, ppx = ((x0 - p.pad) >> 1) + pc;  // (arithmetic shifts: floor for the -1 / -2 of the first tiles)
        const bool item = tid < 2 * PR * PCW;
        const bool ok = item && ppy >= 0 && ppy < (p.H >> 1) && ppx >= 0 && ppx < st_w;
        const int pidx = ppy * st_w + ppx;
        voff[0] = ok ? (unsigned)(oct * 8 * st_plane + pidx) * 4u : 0x80000000u;
        vcode = ok ? (unsigned)(oct * st_plane + pidx) * 8u : 0x80000000u;
        const int r0 = 2 * ppy - (y0 - p.pad), c0 = 2 * ppx - (x0 - p.pad);  // patch position of the window's first corner (-1 ... )
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + (q >> 1), c = c0 + (q & 1);
            if (item && r >= 0 && r < XW_ROWS + 2 && c >= 0 && c < XW_PC) inmask |= 1u << q;
        }
        lds_w[0] = (unsigned)(oct * XW_PLANE + (r0 * XW_PC + c0) * 16);  // (of corner 0; only ever used plus a corner's offset, for corners inside)
        voff[1] = voff[2] = 0x80000000u;
        lds_w[1] = lds_w[2] = 0;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + 256 * k;
            const int oct = idx >= XW_NPOS ? 1 : 0;
            const int pos = idx - oct * XW_NPOS;
            const int r = pos / XW_PC, c = pos - r * XW_PC;
            const int iy = y0 + r - p.pad, ix = x0 + c - p.pad;
            const bool ok = idx < XW_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            voff[k] = ok ? (unsigned)(oct * 8 * in_plane + iy * p.W + ix) * 4u : 0x80000000u;
            lds_w[k] = (unsigned)(oct * XW_PLANE + pos * 16);
        }
    }
    // buffer resource over 16 planes from the chunk's first one: the range check sees the vector offset only
    const unsigned range = (unsigned)in_plane * 64u;    // (planes of the layer's own size: the fused Gram backward's F)
    const unsigned range_x = (unsigned)st_plane * 64u;  // planes of the staged array
    constexpr int NI = UNPOOL ? 1 : 3;  // staging items per thread
    float rp[NI][8];
    u32x2 cd = {0u, 0u};  // UNPOOL: the item's decision bytes
    unsigned sel2[4];     // UNPOOL: the decisions two per register (16-bit lanes, channels 2 i and 2 i + 1), ReLU bit masked as asked
    u32x4 hl[3][2];  // the split chunk waiting for its LDS write: [item][part]  (UNPOOL: [0], [1] = corners 0 and 1, [2] = the unmasked split)
    // The chunk list of a workgroup: the layer's input in 16-channel chunks, then (fused Gram backward) the 16-channel chunks
    // of the layer's OUTPUT-side feature map F = omask, which meet the one-tap bank of D.
    const int nmain = p.Cin / 16;
    const int n2 = p.dbank ? p.Cout / 16 : 0;
    const float* __restrict__ fin = p.dbank ? p.omask + (int64_t)n * p.Cout * in_plane : nullptr;  // (same plane size: checked on the host)
    auto load_patch = [&](int ch) {
        asm volatile("" : "+s"(ch));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (int64_t)ch * 16 * st_plane), 0, range_x, 0x00020000);
        if constexpr (UNPOOL) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xcodes + (int64_t)ch * 16 * st_plane), 0,
                                                                                range_x >> 2, 0x00020000);
            cd = __builtin_amdgcn_raw_buffer_load_b64(rc, vcode, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < NI; ++k)
                rp[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[k], c * st_plane * 4, 0));
    };
    auto publish_max = [&]() {
        if constexpr (UNPOOL) {
            // A value counts (for the chunk's scale, and at all) where its byte names a corner that is a patch position - bit 2 of the
            // byte, kept by code_mask = 7, names none.  What another tile's patch holds of this window is that tile's business.
            const unsigned m2 = code_mask * 0x00010001u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sel2[i] = __builtin_amdgcn_perm(0u, cd[i >> 1], (i & 1) ? 0x0c030c02u : 0x0c010c00u) & m2;
                const unsigned s0 = sel2[i] & 0xffffu, s1 = sel2[i] >> 16;
                rp[0][2 * i] = ((inmask >> s0) & 1u) ? rp[0][2 * i] : 0.f;
                rp[0][2 * i + 1] = ((inmask >> s1) & 1u) ? rp[0][2 * i + 1] : 0.f;
            }
        }
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NI; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(rp[k][c]));
        m = wave_max_nonneg(m);
        if (lane == 0) Ml[UNPOOL ? wv : wave] = m;
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
    // split of one staged item: hl[k][0] <- packed high parts, hl[k][1] <- packed low parts (bit patterns)
    // UNPOOL: step 0 splits the item into hl[2]; step 1 cuts the vectors of corners 0 and 1 out of it (corners 2 and 3: store_patch, so
    // that no more registers wait for the write than in the plain kernel)
    auto corner_of = [&](int q, int part, int i) {  // dword i of corner q's vector: the halves of the channels whose byte names q
        const unsigned t = sel2[i] ^ ((unsigned)q * 0x00010001u);                           // 16-bit lane == 0 where it does
        const unsigned keep = ((t & 0xffffu) ? 0u : 0xffffu) | ((t >> 16) ? 0u : 0xffff0000u);
        return hl[2][part][i] & keep;
    };
    auto split_item = [&](int k) {
        if constexpr (UNPOOL) {
            if (k == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hl[q][0][i] = corner_of(q, 0, i);
                        hl[q][1][i] = corner_of(q, 1, i);
                    }
            }
            if (k > 0) return;
        }
        const int dst = UNPOOL ? 2 : k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = rp[k][2 * q] * sx, v1 = rp[k][2 * q + 1] * sx;
            const unsigned H = xw_cvt_pk_f16(v0, v1);
            const unsigned Lo = xw_cvt_pk_f16(v0 - xw_f16_lo(H), v1 - xw_f16_hi(H));
            hl[dst][0][q] = H;
            hl[dst][1][q] = Lo;
        }
    };
    auto store_patch = [&]() {
        if constexpr (UNPOOL) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((inmask >> q) & 1u) {
                    const unsigned dst = lds_w[0] + (unsigned)(((q >> 1) * XW_PC + (q & 1)) * 16);
                    u32x4 h, l;
                    if (q < 2) {
                        h = hl[q][0];
                        l = hl[q][1];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            h[i] = corner_of(q, 0, i);
                            l[i] = corner_of(q, 1, i);
                        }
                    }
                    *reinterpret_cast<u32x4*>(Pl + dst) = h;
                    *reinterpret_cast<u32x4*>(Pl + 2 * XW_PLANE + dst) = l;
                }
            return;
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (k == 2 && tid >= XW_ITEMS - 512) break;
            *reinterpret_cast<u32x4*>(Pl + lds_w[k]) = hl[k][0];
            *reinterpret_cast<u32x4*>(Pl + 2 * XW_PLANE + lds_w[k]) = hl[k][1];
        }
    };

    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    // Filter slice of a chunk = 36 planes of 1 KiB in LDS order; wave w streams planes w, w + 4, ... (9 LDS-DMA instructions)
    const unsigned lane16 = lane * 16;
    // planes [first, first + 4 count) of chunk ch, one per wave and step: taps 0-4 are planes 0-19 (5 per wave), taps 5-8 planes 20-35
    auto dma_filters = [&](int ch, int first, int count) {
        const unsigned char* src = bank + ((int64_t)ch * ntile + cotile) * XW_W_BYTES;
#pragma unroll
        for (int i = 0; i < count; ++i) {
            const int q = first + wv + 4 * i;
            const unsigned char* g = src + q * 1024;
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(Wl + q * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(g)
                         : "memory");
        }
    };

    const unsigned char* __restrict__ dbank = reinterpret_cast<const unsigned char*>(p.dbank);
    const float d_inv_scale = p.dbank ? p.dinv[n] : 0.f;

    // fragment byte offsets of this lane: patch (row 2 wave + row + ky, col j + kx, octet = lane half), filters (co = j)
    const int b_base = half * XW_PLANE + ((2 * wave) * XW_PC + j) * 16;
    const int a_base = half * 1024 + j * 16;

    f32x16 acc[2][2], master[2][2];  // [32-channel half of the tile][row]
    {
        const bool with_bias = p.bias != nullptr && p.ksplit <= 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float b0 = 0.f;
                if (with_bias) b0 = p.bias[min(co, p.Cout - 1)];
                master[t][0][r] = b0;
                master[t][1][r] = b0;
                acc[t][0][r] = 0.f;
                acc[t][1][r] = 0.f;
            }
    }

    // One tap = 4 filter + 4 patch fragments (ds_read_b128) and 12 MFMAs, issued as two half-steps of 6 (one per 32-channel
    // half of the tile).  Fragments are requested one to two half-steps (192-384 matrix cycles) before their MFMAs - two
    // patch sets, one filter set per half, sched_barrier keeps the order - so LDS latency hides behind the matrix pipe
    // instead of stalling the wave several times per tap.
    typedef f16x8 AFrag[2];     // [part] of one 32-channel half
    typedef f16x8 BFrag[2][2];  // [row][part]
    auto load_a = [&](AFrag& a, int tap, int t) {
#pragma unroll
        for (int part = 0; part < 2; ++part)
            a[part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 2048 + t * 512);
    };
    auto load_b = [&](BFrag& b, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                b[row][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 2 * XW_PLANE + ((row + ky) * XW_PC + kx) * 16);
    };
    auto mfma_half = [&](const AFrag& a, const BFrag& b, int t) {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[row][0], acc[t][row], 0, 0, 0);  // smallest terms first
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[row][1], acc[t][row], 0, 0, 0);
            acc[t][row] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[row][0], acc[t][row], 0, 0, 0);
        }
    };
    AFrag a0, a1;
    BFrag bx, by;
#define XW_FENCE() __builtin_amdgcn_sched_barrier(0)
// tap TAP with patch set BCUR; requests patch set BNXT and the first filter half of tap TAP + 1 (when HAS_NEXT); EXTRA0 / EXTRA1
// are scheduled among the MFMAs of the two half-steps
#define XW_TAP(TAP, BCUR, BNXT, HAS_NEXT, EXTRA0, EXTRA1) \
    do {                                                  \
        load_a(a1, TAP, 1);                               \
        if (HAS_NEXT) load_b(BNXT, (TAP) + 1);            \
        XW_FENCE();                                       \
        mfma_half(a0, BCUR, 0);                           \
        EXTRA0;                                           \
        XW_FENCE();                                       \
        if (HAS_NEXT) load_a(a0, (TAP) + 1, 0);           \
        XW_FENCE();                                       \
        mfma_half(a1, BCUR, 1);                           \
        EXTRA1;                                           \
        XW_FENCE();                                       \
    } while (0)

    // chunk c:  [buffer loads of patch(c+1)]  taps 0-4, max(c+1) -> LDS | XM | DMA filter planes of taps 0-4 (c+1), scale(c+1),
    //           taps 5-8 with the split of patch(c+1) between them | X1 | DMA planes of taps 5-8, write patch(c+1), fold acc x inv(c),
    //           wait | X2 | next chunk
    const int nchunks_all = nmain + n2;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);
    float inv_cur = 0.f, inv_next = 0.f;  // un-scaling factor of the chunk whose products are accumulating / of the next one
    if (ch_begin < min(nchunks, nmain)) {  // (a split that starts inside F, or past the end of the list, skips the 3x3 part)
        dma_filters(ch_begin, 0, 9);
        load_patch(ch_begin);
        publish_max();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        inv_cur = chunk_scale() * w_inv_scale;
        inv_next = inv_cur;
#pragma unroll
        for (int k = 0; k < 3; ++k) split_item(k);
        store_patch();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // Two workgroups share a CU (one wave each per SIMD).  Launched together they run in lockstep - both in their matrix
    // phases (each at half rate), then both staging (the pipe idle): measured 2 x 3456 + 4400 cycles per chunk.  The
    // workgroups of the first round that sit in an odd slot of their CU start half a period late, so that one stages while
    // the other multiplies; later rounds inherit the offset (partners then finish at different times).
    if (p.stagger > 0) {
        const int tg_slot = (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4) & 1);  // hwreg(HW_REG_HW_ID, 16, 4): TG_ID
        const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (tg_slot && flat < 512)
            for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(8);  // 8 x 64 cycles each
    }

#ifdef XW_STAMP
    if (lane == 0 && p.mask) {  // which CU / SIMD this wave runs on, shader clock and 100 MHz clock at loop start: slot 63
        float* d_ = const_cast<float*>(p.mask) + ((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 64 + 63) * 8;
        d_[0] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4));
        d_[1] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[2] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif
    const int main_end = min(nchunks, nmain);
    const int d_begin = max(ch_begin, nmain);  // first chunk of F (if any: d_begin < nchunks)
    for (int ch = ch_begin; ch < main_end; ++ch) {
        const bool more = ch + 1 < main_end;
        XW_MARK(0);
        load_b(bx, 0);
        load_a(a0, 0, 0);
        if (more) load_patch(ch + 1);
        XW_FENCE();
        XW_TAP(0, bx, by, true, (void)0, (void)0);
        XW_TAP(1, by, bx, true, (void)0, (void)0);
        XW_TAP(2, bx, by, true, (void)0, (void)0);
        XW_TAP(3, by, bx, true, (void)0, (void)0);
        XW_TAP(4, bx, by, true, (void)0, if (more) publish_max());  // requests tap 5: patch and planes 20-35 stay across XM
        XW_MARK(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        XW_MARK(2);
        __builtin_amdgcn_s_barrier();  // XM: the filter planes of taps 0-4 are free; the maxima of the next chunk are visible
        XW_MARK(3);
        if (more) {
            dma_filters(ch + 1, 0, 5);
            inv_next = chunk_scale() * w_inv_scale;
        }
        XW_FENCE();
        XW_TAP(5, by, bx, true, if (more) split_item(0), (void)0);
        XW_TAP(6, bx, by, true, if (more) split_item(1), (void)0);
        XW_TAP(7, by, bx, true, if (more) split_item(2), (void)0);
        XW_TAP(8, bx, by, false, (void)0, (void)0);
        XW_MARK(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X1: every wave is done reading the patch and the remaining filter planes
        XW_MARK(5);
        if (more) {
            dma_filters(ch + 1, 20, 4);
            store_patch();
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[t][row][r] = fmaf(acc[t][row][r], inv_cur, master[t][row][r]);  // fold + un-scale (power of two: exact)
                    acc[t][row][r] = 0.f;
                }
        inv_cur = inv_next;
        XW_MARK(6);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            XW_MARK(7);
            __builtin_amdgcn_s_barrier();  // X2: patch and filters of the next chunk are in LDS
        }
    }

    // Fused Gram backward: the 16-channel chunks of F against the matching columns of D - one tap, 12 MFMAs per chunk and wave.
    // Nothing here goes through LDS: a lane's MFMA operands are exactly what it can load itself - its pixel column of its two
    // rows for 8 channels (B operand: 16 coalesced dword loads), its output channel's 8 values of D (A operand: one 16-byte
    // load per part and channel half from the packed bank) - and the power-of-two scale is per WAVE (its own maximum, its own
    // accumulators).  No barrier, no cross-wave exchange: the waves drift apart and hide each other's load latency, the loads
    // of chunk c + 1 are in flight while chunk c is multiplied.
    if constexpr (!POOL) if (d_begin < nchunks) {
        __builtin_amdgcn_s_barrier();  // (every wave is done with the LDS fragments of the last 3x3 chunk - nothing below needs LDS,
                                       //  but the stamps / a later reader of this code should not have to wonder)
        unsigned voff_f[2];
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int oy = y0 + 2 * wave + row, oxx = x0 + j;
            voff_f[row] = (oy < p.H && oxx < p.W) ? (unsigned)(half * 8 * in_plane + oy * p.W + oxx) * 4u : 0x80000000u;
        }
        struct DChunk {
            float f[2][8];   // [row][channel of this lane's octet]
            u32x4 a[2][2];   // [part][32-channel half of the tile]: 8 fp16 of D[co][c0 + 8 half ..]
        };
        auto request = [&](DChunk& d, int c2) {
            asm volatile("" : "+s"(c2));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fin + (int64_t)c2 * 16 * in_plane), 0, range, 0x00020000);
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int row = 0; row < 2; ++row)
                    d.f[row][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff_f[row], c * in_plane * 4, 0));
            const unsigned char* g = dbank + (((int64_t)n * n2 + c2) * ntile + cotile) * 4096 + half * 1024 + j * 16;
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int t = 0; t < 2; ++t) d.a[part][t] = *reinterpret_cast<const u32x4*>(g + part * 2048 + t * 512);
        };
        auto multiply = [&](DChunk& d) {
            float m = 0.f;
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(d.f[row][c]));
            m = wave_max_nonneg(m);
            int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
            e = m > 0.f ? max(e, -100) : 11;
            const float s = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
            const float inv = __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23) * d_inv_scale;
            BFrag b;
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                u32x4 Hh, Ll;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v0 = d.f[row][2 * q] * s, v1 = d.f[row][2 * q + 1] * s;
                    const unsigned H = xw_cvt_pk_f16(v0, v1);
                    Hh[q] = H;
                    Ll[q] = xw_cvt_pk_f16(v0 - xw_f16_lo(H), v1 - xw_f16_hi(H));
                }
                b[row][0] = __builtin_bit_cast(f16x8, Hh);
                b[row][1] = __builtin_bit_cast(f16x8, Ll);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                AFrag a;
                a[0] = __builtin_bit_cast(f16x8, d.a[0][t]);
                a[1] = __builtin_bit_cast(f16x8, d.a[1][t]);
                mfma_half(a, b, t);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int row = 0; row < 2; ++row)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        master[t][row][q] = fmaf(acc[t][row][q], inv, master[t][row][q]);
                        acc[t][row][q] = 0.f;
                    }
        };
        DChunk da, db;
        request(da, d_begin - nmain);
        for (int ch = d_begin; ch < nchunks; ch += 2) {
            if (ch + 1 < nchunks) request(db, ch + 1 - nmain);
            multiply(da);
            if (ch + 1 < nchunks) {
                if (ch + 2 < nchunks) request(da, ch + 2 - nmain);
                multiply(db);
            }
        }
    }

#ifdef XW_STAMP
    if (lane == 0 && p.mask) {
        float* d_ = const_cast<float*>(p.mask) + ((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 64 + 63) * 8;
        d_[3] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[4] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif
    // Split channel loop finished INSIDE the launch (round 6; conv_x3q.hip has the description): every split's workgroup leaves its partial
    // sums in its slab (lane-linear, sixteen 16-byte vectors per thread, written through and drained) and draws a ticket; the last arriver
    // of a tile adds the slabs in split order - conv_splitk_finish_kernel's additions - then the bias, and runs the one-pass epilogue.
    // (Nothing of this phase lives across the K loop: tile number and split index wait in LDS, the arguments are read again.)
    bool final_sums = p.ksplit <= 1;
    const ConvArgs* kp = xw_kernel_arguments();
    asm volatile("" : "+s"(kp));
    if (kp->ksplit > 1 && kp->arrive != nullptr) {
        const int ksplit = kp->ksplit;
        const int unit = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(Ml)[5]);
        const int split = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(Ml)[6]);
        unsigned* const arrive = kp->arrive + unit;
        const float* const bias = kp->bias;
        constexpr unsigned SLAB = XW_COT * XW_ROWS * 32 * 4;  // 64 KiB
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(kp->ws) + (int64_t)unit * ksplit * SLAB, 0,
                                                                            (unsigned)ksplit * SLAB, 0x00020000);
        const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x16& m = master[q >> 3][(q >> 2) & 1];
            const f32x4 v = {m[4 * (q & 3)], m[4 * (q & 3) + 1], m[4 * (q & 3) + 2], m[4 * (q & 3) + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, toff, (unsigned)split * SLAB + q * 4096u, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned* tick = reinterpret_cast<unsigned*>(Ml) + 4;
        if (tid == 0) *tick = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if ((int)*tick != ksplit - 1) return;
        if (tid == 0) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][row][r] = 0.f;
        for (int k = 0; k < ksplit; ++k) {
            if (k == split) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int row = 0; row < 2; ++row) acc[t][row] += master[t][row];
            } else {
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += 8) {
                    u32x4 tq[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) tq[q] = __builtin_amdgcn_raw_buffer_load_b128(srs, toff, (unsigned)k * SLAB + (q0 + q) * 4096u, 16);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x4 v = __builtin_bit_cast(f32x4, tq[q]);
                        f32x16& a = acc[(q0 + q) >> 3][((q0 + q) >> 2) & 1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[4 * ((q0 + q) & 3) + e] += v[e];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float b0 = bias != nullptr ? bias[min(co, kp->Cout - 1)] : 0.f;
                master[t][0][r] = acc[t][0][r] + b0;
                master[t][1][r] = acc[t][1][r] + b0;
            }
        final_sums = true;
    }
    // epilogue: lane holds pixel column j of rows y0 + 2 wave + {0, 1}; register r is output channel (r&3)+8*(r>>2)+4*half
    if constexpr (POOL) if (final_sums) {
        // ReLU + the 2x2 / 2 max pool behind it: a wave's two rows and neighbouring lanes are exactly the windows, so the
        // full-size activation never goes to memory - only the pooled map and one decision byte per window (what
        // pool2x2_fwd_codes_kernel leaves: position of the first maximum in scan order, bit 2 = the maximum is <= 0; bytes laid out
        // [octet of channels][pooled pixel][8 channels], Cout % 8 == 0).
        const int PW = p.OW >> 1;
        const int64_t pplane = (int64_t)(p.OH >> 1) * PW;
        float* __restrict__ py = p.y + (int64_t)n * p.Cout * pplane;
        unsigned char* __restrict__ pc = p.pool_codes + (int64_t)n * p.Cout * pplane;
        const int oy = y0 + 2 * wave, oxx = x0 + j;
        const bool store = (lane & 1) == 0 && oy + 1 < p.OH && oxx + 1 < p.OW;  // (a window is inside or outside; an odd plane's last row / column has none)
        const int64_t ppix = (int64_t)(oy >> 1) * PW + (oxx >> 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // four consecutive channels per step: their decision bytes are one dword of the octet-interleaved layout
                const int cq = co0 + t * 32 + 8 * q + 4 * half;
                unsigned pk = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    float a = master[t][0][r], c = master[t][1][r];
                    a = a > 0.f ? a : 0.f;
                    c = c > 0.f ? c : 0.f;
                    const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x101, 0xf, 0xf, false));  // row_shl:1
                    const float d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x101, 0xf, 0xf, false));
                    float m = a;
                    unsigned arg = 0;
                    if (b > m) { m = b; arg = 1; }
                    if (c > m) { m = c; arg = 2; }
                    if (d > m) { m = d; arg = 3; }
                    pk |= (arg | (m > 0.f ? 0u : 4u)) << (8 * i);
                    if (store && cq < p.Cout) py[(int64_t)(cq + i) * pplane + ppix] = m;
                }
                if (store && cq < p.Cout) *reinterpret_cast<unsigned*>(pc + ((int64_t)(cq >> 3) * pplane + ppix) * 8 + 4 * half) = pk;
            }
        return;
    }
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
    const int ox = x0 + j;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const int oy = y0 + 2 * wave + row;
        const bool pvalid = oy < p.OH && ox < p.OW;
        const int64_t opix = (int64_t)oy * p.OW + ox;
        if (!final_sums) {  // split-K: un-scaled partial sums, finished by conv_splitk_finish_kernel in split order
            float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * out_plane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pvalid && co < p.Cout) wsp[(int64_t)co * out_plane + opix] = master[t][row][r];
                }
            continue;
        }
        if (!pvalid) continue;
        const bool full = co0 + XW_COT <= p.Cout;
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
                float v = master[t][row][r];
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

// split-K over 16-channel chunks when the grid leaves most of the 512 workgroup slots (2 per CU) empty
static int x3w_choose_split(const ConvArgs& a, int n) {
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + XW_ROWS - 1) / XW_ROWS) * ((a.Cout + XW_COT - 1) / XW_COT) * split_batch_hint();
    (void)n;  // The policy does not look at THIS launch's batch size but at the number of frames the caller plans to evaluate per
              // launch (maua_set_split_batch_hint, 1 by default): vid_img's frames are then computed with the same summation
              // order whether they run one at a time or sixteen together - bit-identical results whatever the grouping.
    const int nchunks = a.Cin / 16;
    const int forced = (int)tuning("x3w_ks", 0);  // experiments: splits every launch k ways (when the layer has the chunks)
    if (forced > 0) return forced <= nchunks / 2 ? forced : (nchunks >= 4 ? nchunks / 2 : 1);
    if (wgs >= 2048 || nchunks < 4) return 1;
    const double out_mb = (double)split_batch_hint() * a.Cout * a.OH * a.OW * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 2; ++ks) {
        const double rounds = (double)((wgs * ks + 511) / 512);
        double cost = rounds * ((double)((nchunks + ks - 1) / ks) + 1.0) * 4.0;  // ~4 us per 16-channel chunk of a full CU
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

bool conv_x3w_supports(const ConvArgs& a) {
    return a.Cin % 16 == 0 && (int64_t)a.H * a.W <= (1ll << 25) && a.pad >= 0 && a.pad <= 2;
}

#ifdef XW_STAMP
static float* g_xw_stamp = nullptr;
extern "C" void maua_xw_set_stamp_buffer(float* buf) { g_xw_stamp = buf; }
#endif

// In-launch finish of a split channel loop (conv_x3q.hip): one 64 KiB slab per (image, channel tile, pixel tile, split), whole tiles.
static size_t x3w_in_launch_bytes(const ConvArgs& a, int n, int ks) {
    const int64_t tiles = (int64_t)((a.OW + 31) / 32) * ((a.OH + XW_ROWS - 1) / XW_ROWS), cot = (a.Cout + XW_COT - 1) / XW_COT;
    return (size_t)n * cot * tiles * ks * (XW_COT * XW_ROWS * 32 * 4);
}
static bool x3w_finish_in_launch(const ConvArgs& a, int n, int ks) {
    const int64_t units = (int64_t)n * ((a.Cout + XW_COT - 1) / XW_COT) * ((a.OW + 31) / 32) * ((a.OH + XW_ROWS - 1) / XW_ROWS);
    return ks > 1 && a.arrive != nullptr && ks <= (int)tuning("finish_in_launch_max_ks", 4) && units <= ARRIVE_COUNTERS;
}

int conv_x3w_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
#ifdef XW_STAMP
    p.mask = g_xw_stamp;
#endif
    p.tiles_x = (a.OW + 31) / 32;
    p.cot_inner = tuning("cot_inner", 0) != 0 ? 1 : 0;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + XW_ROWS - 1) / XW_ROWS);
    const int ks = a.ws ? x3w_choose_split(a, n) : 1;
    p.ksplit = ks;
    const int64_t cot = (a.Cout + XW_COT - 1) / XW_COT, per_xcd = (tiles + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8), (unsigned)cot, (unsigned)(n * ks));
    const bool whole = ks == 1 || x3w_finish_in_launch(a, n, ks);  // the launch's epilogue holds complete sums (one pass, or the last arriver's)
    if (!whole) p.arrive = nullptr;
    const bool acc = whole && a.accumulate != 0, om = whole && a.omask != nullptr;
    const float w_inv = 1.f / w_scale;
    {
        p.stagger = (int)tuning("x3w_stagger", 7);
    }
    if (a.pool_codes && whole) hipLaunchKernelGGL((conv_x3w_kernel<false, false, true>), grid, dim3(256), 0, stream, p, w_inv);
    else if (a.in_codes && om) hipLaunchKernelGGL((conv_x3w_kernel<false, true, false, true>), grid, dim3(256), 0, stream, p, w_inv);
    else if (a.in_codes) hipLaunchKernelGGL((conv_x3w_kernel<false, false, false, true>), grid, dim3(256), 0, stream, p, w_inv);
    else if (acc && om) hipLaunchKernelGGL((conv_x3w_kernel<true, true>), grid, dim3(256), 0, stream, p, w_inv);
    else if (acc) hipLaunchKernelGGL((conv_x3w_kernel<true, false>), grid, dim3(256), 0, stream, p, w_inv);
    else if (om) hipLaunchKernelGGL((conv_x3w_kernel<false, true>), grid, dim3(256), 0, stream, p, w_inv);
    else hipLaunchKernelGGL((conv_x3w_kernel<false, false>), grid, dim3(256), 0, stream, p, w_inv);
    int rc = check_launch("conv_x3w_kernel");
    if (rc || whole) return rc;
    // (a split channel loop leaves partial sums: the ReLU + pool of a pooling launch then happen in the pass that adds them)
    return a.pool_codes ? conv_splitk_finish_pool(a, n, ks, stream) : conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_x3w_bank_bytes(int cout_produced, int cin_consumed) {
    if (cout_produced <= 0 || cin_consumed <= 0 || cout_produced > (1 << 20) || cin_consumed > (1 << 20)) return 0;
    const size_t nchunk = (cin_consumed + 15) / 16, ntile = (cout_produced + XW_COT - 1) / XW_COT;
    return nchunk * ntile * XW_W_BYTES;
}

int maua_conv_pack_filters_x3w(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                               maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20) && w_scale > 0.f, MAUA_E_INVAL,
                 "conv_pack_filters_x3w: bad args");
    if (bank_fwd) {
        hipLaunchKernelGGL(pack_x3w_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_fwd, cout,
                           cin, 0, w_scale);
        int rc = check_launch("pack_x3w_kernel");
        if (rc) return rc;
    }
    if (bank_bwd) {
        hipLaunchKernelGGL(pack_x3w_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_bwd, cout,
                           cin, 1, w_scale);
        return check_launch("pack_x3w_kernel");
    }
    return MAUA_OK;
}

size_t maua_conv_x3w_dmat_bank_bytes(int c) {
    if (c <= 0 || c % 16 != 0 || c > (1 << 14)) return 0;
    return (size_t)(c / 16) * ((c + XW_COT - 1) / XW_COT) * 4096;
}

int maua_conv_pack_dmat_x3w(const float* dmat, int c, void* bank, float* inv_scale_out, maua_stream_t stream) {
    MAUA_REQUIRE(dmat && bank && inv_scale_out && c > 0 && c % 16 == 0 && c <= (1 << 14), MAUA_E_INVAL,
                 "conv_pack_dmat_x3w: needs a C x C matrix with C %% 16 == 0");
    const int64_t groups = (int64_t)(c / 16) * ((c + XW_COT - 1) / XW_COT) * 2 * XW_COT;
    hipLaunchKernelGGL(pack_dmat_x3w_kernel, dim3((unsigned)((groups + 4095) / 4096)), dim3(1024), 0, (hipStream_t)stream, dmat, c,
                       (unsigned short*)bank, inv_scale_out);
    return check_launch("pack_dmat_x3w_kernel");
}

int maua_conv_pack_dmat_x3w_batch(int count, const float* const* dmats, const int* cs, void* const* banks, float* const* inv_scales_out,
                                  maua_stream_t stream) {
    MAUA_REQUIRE(count > 0 && count <= 4 && dmats && cs && banks && inv_scales_out, MAUA_E_INVAL, "conv_pack_dmat_x3w_batch: bad args (at most 4 layers)");
    DmatPackBatch b{};
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        MAUA_REQUIRE(dmats[i] && banks[i] && inv_scales_out[i] && cs[i] > 0 && cs[i] % 16 == 0 && cs[i] <= (1 << 14), MAUA_E_INVAL,
                     "conv_pack_dmat_x3w_batch: layer %d needs a C x C matrix with C %% 16 == 0", i);
        b.d[i] = dmats[i];
        b.bank[i] = (unsigned short*)banks[i];
        b.inv[i] = inv_scales_out[i];
        b.C[i] = cs[i];
        const int64_t groups = (int64_t)(cs[i] / 16) * ((cs[i] + XW_COT - 1) / XW_COT) * 2 * XW_COT;
        most = groups > most ? groups : most;
    }
    hipLaunchKernelGGL(pack_dmat_x3w_batch_kernel, dim3((unsigned)((most + 4095) / 4096), (unsigned)count), dim3(1024), 0, (hipStream_t)stream, b);
    return check_launch("pack_dmat_x3w_batch_kernel");
}

int maua_conv_x3w_supported(int cin, int h, int w, int pad) {
    ConvArgs a{};
    a.Cin = cin;
    a.H = h;
    a.W = w;
    a.pad = pad;
    return conv_dims_ok(1, cin, h, w, 1, pad) && conv_x3w_supports(a) ? 1 : 0;
}

size_t maua_conv_x3w_workspace_bytes(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const int ks = x3w_choose_split(a, n);
    if (ks <= 1) return 0;
    const size_t two_launches = (size_t)n * ks * cout * a.OH * a.OW * sizeof(float);
    const size_t in_launch = x3w_in_launch_bytes(a, n, ks);  // (whole tiles; a caller that arms its workspace gets this form)
    return in_launch > two_launches ? in_launch : two_launches;
}

static int conv3x3_x3w_entry(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                             int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, const void* dbank,
                             const float* dinv, void* workspace, size_t workspace_bytes, maua_stream_t stream,
                             const unsigned char* in_codes = nullptr, int in_code_mask = 0) {
    MAUA_REQUIRE(x && bank && y && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3w: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3w: bad dims");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_x3w: input smaller than the filter");
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
    MAUA_REQUIRE(conv_x3w_supports(a), MAUA_E_UNSUPPORTED, "conv3x3_x3w: needs cin %% 16 == 0 and a plane of at most 2^25 pixels");
    if (dbank) {
        MAUA_REQUIRE(dinv && out_relu_mask, MAUA_E_INVAL, "conv3x3_x3w_gram: needs the feature map and the bank's inverse scale");
        MAUA_REQUIRE(cout % 16 == 0 && a.OH == h && a.OW == w, MAUA_E_UNSUPPORTED,
                     "conv3x3_x3w_gram: needs cout %% 16 == 0 and an output plane of the input's size");
        a.dbank = dbank;
        a.dinv = dinv;
    }
    if (in_codes) {
        MAUA_REQUIRE(h >= 2 && w >= 2 && !accumulate, MAUA_E_UNSUPPORTED, "conv3x3_x3w_unpool: needs an input plane of 2 x 2 and more, no accumulation");
        a.in_codes = in_codes;
        a.in_code_mask = in_code_mask;
    }
    a.ws = (workspace && workspace_bytes >= maua_conv_x3w_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    a.arrive = a.ws ? armed_counters(workspace) : nullptr;  // (the calling thread armed this workspace: small splits finish inside the launch)
    return conv_x3w_launch(a, n, w_scale, (hipStream_t)stream);
}

int maua_conv3x3_x3w(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream) {
    return conv3x3_x3w_entry(x, bank, w_scale, bias, out_relu_mask, y, n, cin, h, w, cout, pad, relu, accumulate, nullptr, nullptr,
                             workspace, workspace_bytes, stream);
}

int maua_conv_x3w_split(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    return x3w_choose_split(a, n);
}

int maua_conv3x3_x3w_relu_pool(const float* x, const void* bank, float w_scale, const float* bias, float* pooled,
                               unsigned char* codes, int n, int cin, int h, int w, int cout, int pad, void* workspace,
                               size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && pooled && codes && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3w_relu_pool: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3w_relu_pool: bad dims");
    ConvArgs a{};
    a.x = x;
    a.w6 = bank;
    a.bias = bias;
    a.y = pooled;
    a.pool_codes = codes;
    a.Cin = cin;
    a.H = h;
    a.W = w;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    a.pad = pad;
    a.relu = 1;
    MAUA_REQUIRE(a.OH >= 2 && a.OW >= 2 && cout % 8 == 0 && conv_x3w_supports(a), MAUA_E_UNSUPPORTED,
                 "conv3x3_x3w_relu_pool: needs an output plane of 2 x 2 and more, cin %% 16 == 0, cout %% 8 == 0");
    // without a workspace: one pass over the channels, the epilogue holds complete sums and pools them itself
    a.ws = (workspace && workspace_bytes >= maua_conv_x3w_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    a.arrive = a.ws ? armed_counters(workspace) : nullptr;  // (the calling thread armed this workspace: small splits finish inside the launch)
    return conv_x3w_launch(a, n, w_scale, (hipStream_t)stream);
}

int maua_conv3x3_x3w_gram(const float* x, const void* bank, float w_scale, const float* feature_map, const void* dmat_bank,
                          const float* dmat_inv_scale, float* y, int n, int cin, int h, int w, int cout, int pad, int accumulate,
                          void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(dmat_bank, MAUA_E_INVAL, "conv3x3_x3w_gram: null bank");
    return conv3x3_x3w_entry(x, bank, w_scale, nullptr, feature_map, y, n, cin, h, w, cout, pad, 0, accumulate, dmat_bank,
                             dmat_inv_scale, workspace, workspace_bytes, stream);
}

int maua_conv3x3_x3w_unpool(const float* pooled_x, const unsigned char* codes, int honour_relu_bit, const void* bank, float w_scale,
                            const float* out_relu_mask, const void* dmat_bank, const float* dmat_inv_scale, float* y, int n, int cin,
                            int h, int w, int cout, int pad, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(codes, MAUA_E_INVAL, "conv3x3_x3w_unpool: null decision bytes");
    return conv3x3_x3w_entry(pooled_x, bank, w_scale, nullptr, out_relu_mask, y, n, cin, h, w, cout, pad, 0, 0, dmat_bank, dmat_inv_scale,
                             workspace, workspace_bytes, stream, codes, honour_relu_bit ? 7 : 3);
}

}  // extern "C"
