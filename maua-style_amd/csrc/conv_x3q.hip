// 3x3 stride-1 convolution in fp16x3 arithmetic (conv_x3.hip / conv_x3w.hip), third structure: 32-channel chunks on
// v_mfma_f32_16x16x32_f16, one workgroup per CU.
//
// conv_x3w.hip is bound by the power governor: its matrix pipe is 70 % busy at the 1.55 GHz the chip holds under
// v_mfma_f32_32x32x16_f16.  tools/mfma_probe/conv_loop_shapes.hip (profiles/probe_r04_loop_shapes.txt)
// measures the K loop alone, every operand re-read from LDS, on random data: the 32x32x16 loop runs at 1.55 GHz, the same
// products on 16x16x32 at 1.88 - 1.96 GHz - 13-15 % more matrix throughput at the power limit.  With K = 32 per instruction
// one k-step is ONE tap x 32 input channels, so the chunk is 32 channels: 72 KiB of filters + 44 KiB of patch no longer fit
// twice into a CU's LDS.  This structure therefore runs ONE workgroup of EIGHT waves per CU (two per SIMD, 256 registers):
//   * workgroup = 64 output channels x (16 rows x 32 px); a wave owns 64 channels x (2 rows x 32 px) = 4 x 4 accumulators of
//     16x16 (+ as many fp32 masters, all in architectural VGPRs: no AGPR copies around the folds).  One tap = 8 filter + 8
//     patch fragments (ds_read_b128) for 48 MFMAs (0.33 reads per 16 matrix cycles, as conv_x3w); twice conv_x3w's pixels per
//     filter byte streamed from L2 (9 LDS-DMA instructions per wave and chunk), 10 % less halo per pixel;
//   * all nine taps of the chunk are resident (LDS-DMA, 72 planes of 1 KiB: taps 0-4 re-fetched behind the mid-chunk barrier,
//     taps 5-8 behind the end-of-chunk barriers): three barriers per 32 channels (conv_x3w: six);
//   * in-wave software pipeline: filter fragments one cout group ahead, patch fragments refreshed IN PLACE behind their last
//     readers during the last cout group of the previous tap, the next chunk's patch requested during taps 0-3, reduced to its
//     maximum in tap 4, split between the MFMAs of taps 6-7, the LDS-DMA instructions one per step (each costs its wave ~100
//     cycles of issue); the folds into the masters sit beside the MFMAs that start the new sums from a zero C operand;
//     XQ_PIPE interleaves every region's loads and vector-ALU work with its MFMAs.
// Measured on the way (profiles/probes_r04.md): a 64 x 128 wave tile on four waves needs 340 architectural VGPRs for masters +
// staging + fragments where a wave has 256 beside its AGPRs; four waves of 64 x 64 (one per SIMD, 8-row tiles, double-buffered
// patch) lose 23 % of the matrix pipe to the issue time of their own LDS-DMA instructions with nothing to cover it (conv4_2 216
// us against conv_x3w's 206); eight waves of 32 x 64 on the same 8-row tile read 0.5 fragments per 16 matrix cycles and re-stream
// the filters per 256 pixels (197 us); this form: 171-187 us (1.22 x conv_x3w on the same box).
// Arithmetic is conv_x3w's: x s = xh + xl (fp16 pair, s = power of two per workgroup and chunk, maximum into [2^11, 2^12)),
// pre-split pre-scaled filter bank, products xh*wl, xl*wh, xh*wh, fp32 masters; sums are folded into the masters twice per
// chunk (after taps 0-4 and 5-8: 480 / 384 products per fold, conv_x3w folds 432).
//
// LDS (150 KiB): patch [part][octet 4][pos 18 x 34 (+12)][8 ch] = 79,872 B (octet planes 256-byte aligned: the 16x16x32 operand
// read - lane = (octet, pixel) - is bank-conflict free), filters [tap][part][octet 4][co 64][8 ch] = 73,728 B.
// hipcc-flags: -fno-slp-vectorize
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace maua {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int XQ_COT = 64;
constexpr int XQ_ROWS = 16, XQ_PR = 18, XQ_PC = 34;
constexpr int XQ_NPOS = XQ_PR * XQ_PC;           // 612
constexpr int XQ_NPOS_PAD = 624;                 // a multiple of 16 positions
constexpr int XQ_PLANE = XQ_NPOS_PAD * 16;       // bytes of one [pos][8 ch] plane
constexpr int XQ_PATCH_BYTES = 8 * XQ_PLANE;     // [part][octet]
constexpr int XQ_TAP_BYTES = 2 * 4 * XQ_COT * 16;  // [part][octet][co][16 B] = 8 planes of 1 KiB
constexpr int XQ_W_BYTES = 9 * XQ_TAP_BYTES;
constexpr int XQ_ITEMS = 4 * XQ_NPOS;            // (octet, position) staging items per chunk: 2448 = 4.8 per thread
constexpr int XQ_NI = 5;
constexpr int XQ_THREADS = 512;
constexpr int XQ_LDS_BYTES = XQ_PATCH_BYTES + XQ_W_BYTES + 64;

__device__ __forceinline__ unsigned xq_cvt_pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float xq_f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float xq_f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// bank[dir][chunk32][cotile][tap][part][octet][co][ch] (fp16, pre-scaled): fwd: co = output channel, ch = input channel,
// tap = ky*3+kx; bwd-data: roles swapped and taps flipped.  Zero padding for channels beyond the tensor.
__global__ void pack_x3q_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int cout, int cin, int backward,
                                float w_scale) {
    const int CO = backward ? cin : cout;
    const int CI = backward ? cout : cin;
    const int nchunk = (CI + 31) / 32, ntile = (CO + XQ_COT - 1) / XQ_COT;
    const int64_t total = (int64_t)nchunk * ntile * 9 * 4 * XQ_COT * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int ch = (int)(r % 8);
        r /= 8;
        const int co = (int)(r % XQ_COT);
        r /= XQ_COT;
        const int oct = (int)(r % 4);
        r /= 4;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * XQ_COT + co, i = chunk * 32 + oct * 8 + ch;
        float v = 0.f;
        if (o < CO && i < CI) {
            if (!backward) v = w[((int64_t)o * cin + i) * 9 + tap];
            else v = w[((int64_t)i * cin + o) * 9 + (8 - tap)];
        }
        v *= w_scale;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        const int64_t base = ((int64_t)chunk * ntile + tile) * (XQ_W_BYTES / 2);
        bank[base + ((((int64_t)tap * 2 + 0) * 4 + oct) * XQ_COT + co) * 8 + ch] = __builtin_bit_cast(unsigned short, h);
        bank[base + ((((int64_t)tap * 2 + 1) * 4 + oct) * XQ_COT + co) * 8 + ch] = __builtin_bit_cast(unsigned short, l);
    }
}

#ifdef XQ_STAMP
#define XQ_MARK(k)                                                                                          \
    do {                                                                                                    \
        if (lane == 0 && p.mask) {                                                                          \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                     \
            const_cast<float*>(p.mask)[((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + (ch - ch_begin)) * 8 + (k)] = \
                __builtin_bit_cast(float, (unsigned)t_);                                                    \
        }                                                                                                   \
    } while (0)
#else
#define XQ_MARK(k) do {} while (0)
#endif

#define XQ_FENCE() __builtin_amdgcn_sched_barrier(0)

// One scheduling region of the K loop = 12 MFMAs + what rides along: per MFMA at most one LDS read, one vector-memory
// instruction and VA vector-ALU instructions, in that order (a lone wave per SIMD hides nothing behind another wave: whatever
// the compiler clusters in front of the MFMAs is a hole in the matrix pipe).
#ifdef XQ_NO_PIPE
#define XQ_PIPE(VA) do {} while (0)
#else
#define XQ_PIPE(VA)                                                   \
    do {                                                              \
        _Pragma("unroll") for (int m_ = 0; m_ < 12; ++m_) {           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x002, VA, 0);       \
        }                                                             \
    } while (0)
#endif

template <bool ACC, bool OM>
__global__ void __launch_bounds__(XQ_THREADS, 2) conv_x3q_kernel(ConvArgs p, float w_inv_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Pl = smem;                    // [part][octet][pos][16 B]
    unsigned char* Wl = smem + XQ_PATCH_BYTES;   // [tap][part][octet][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + XQ_PATCH_BYTES + XQ_W_BYTES);  // per-wave maxima of the chunk being staged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, oct = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    const int cotile = blockIdx.y;
    const int co0 = cotile * XQ_COT;
    const int ntile = gridDim.y;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    // XCD-aware tile order (conv_x6.hip): XCD k owns the k-th contiguous band of tiles
    const int tiles_total = p.tiles_x * ((p.OH + XQ_ROWS - 1) / XQ_ROWS);
    const int per_xcd = (tiles_total + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, tiles_total)) return;  // whole workgroup leaves
    const int x0 = (tile % p.tiles_x) * 32, y0 = (tile / p.tiles_x) * XQ_ROWS;

    // Staging items of this thread: item k = (octet, position) number tid + 256 k.  voff = byte offset of the item's first
    // channel from the chunk's first plane; out-of-image positions and items past the end get an offset beyond the buffer's
    // range, for which a buffer load returns 0 (no selects on the values).
    unsigned voff[XQ_NI], lds_w[XQ_NI];
#pragma unroll
    for (int k = 0; k < XQ_NI; ++k) {
        const int idx = tid + XQ_THREADS * k;
        const int o = idx / XQ_NPOS;
        const int pos = idx - o * XQ_NPOS;
        const int r = pos / XQ_PC, c = pos - r * XQ_PC;
        const int iy = y0 + r - p.pad, ix = x0 + c - p.pad;
        const bool ok = idx < XQ_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        voff[k] = ok ? (unsigned)(o * 8 * in_plane + iy * p.W + ix) * 4u : 0x80000000u;
        lds_w[k] = idx < XQ_ITEMS ? (unsigned)(o * XQ_PLANE + pos * 16) : (unsigned)(XQ_NPOS_PAD - 1) * 16u;  // (items past the end: a padding slot)
    }
    const unsigned range = (unsigned)in_plane * 128u;  // 32 planes from the chunk's first one: the range check sees the vector offset only
    float rp[XQ_NI][8];   // the next chunk's patch: raw values
    auto patch_rsrc = [&](int ch) {
        asm volatile("" : "+s"(ch));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (int64_t)ch * 32 * in_plane), 0, range, 0x00020000);
    };
    auto load_patch_part = [&](const __amdgpu_buffer_rsrc_t rs, int c_lo, int c_hi) {
#pragma unroll
        for (int c = c_lo; c < c_hi; ++c)
#pragma unroll
            for (int k = 0; k < XQ_NI; ++k)
                rp[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[k], c * in_plane * 4, 0));
    };
    auto publish_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < XQ_NI; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(rp[k][c]));
        m = wave_max_nonneg(m);
        if (lane == 0) Ml[wave] = m;
    };
    // scale of the staged chunk: max in [2^11, 2^12) after scaling.  Returns the INVERSE scale, sets `sx`.
    float sx = 1.f;
    auto chunk_scale = [&]() {
        const float m = fmaxf(fmaxf(fmaxf(Ml[0], Ml[1]), fmaxf(Ml[2], Ml[3])), fmaxf(fmaxf(Ml[4], Ml[5]), fmaxf(Ml[6], Ml[7])));
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;  // floor(log2 m) for normal m
        e = m > 0.f ? max(e, -100) : 11;
        sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
        return __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23);
    };
    // split of one staged item into its fp16 pair, in place (rp[k][0..3] <- the packed high parts of channels 2q, 2q + 1, rp[k][4..7] <-
    // the packed low parts): high parts x sx rounded to nearest, low parts = the (exact) remainders rounded to nearest.  x sx is exact
    // (a power of two), so fma(x, sx, -h) is the remainder without an intermediate product: one mixed-precision FMA per half.
    auto split_item = [&](int k) {
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rp[k][c]));  // (pinned between the fences of its step: see the fold)
        unsigned hu[4], lu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = rp[k][2 * q], v1 = rp[k][2 * q + 1];
            const f16x2 h2 = {(_Float16)(v0 * sx), (_Float16)(v1 * sx)};
            const f16x2 l2 = {(_Float16)fmaf(v0, sx, -(float)h2[0]), (_Float16)fmaf(v1, sx, -(float)h2[1])};
            hu[q] = __builtin_bit_cast(unsigned, h2);
            lu[q] = __builtin_bit_cast(unsigned, l2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rp[k][q] = __builtin_bit_cast(float, hu[q]);
            rp[k][4 + q] = __builtin_bit_cast(float, lu[q]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rp[k][c]));
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int k = 0; k < XQ_NI; ++k) {
            const u32x4 h = {__builtin_bit_cast(unsigned, rp[k][0]), __builtin_bit_cast(unsigned, rp[k][1]), __builtin_bit_cast(unsigned, rp[k][2]),
                             __builtin_bit_cast(unsigned, rp[k][3])};
            const u32x4 l = {__builtin_bit_cast(unsigned, rp[k][4]), __builtin_bit_cast(unsigned, rp[k][5]), __builtin_bit_cast(unsigned, rp[k][6]),
                             __builtin_bit_cast(unsigned, rp[k][7])};
            *reinterpret_cast<u32x4*>(Pl + lds_w[k]) = h;   // (items past the end write a padding slot)
            *reinterpret_cast<u32x4*>(Pl + 4 * XQ_PLANE + lds_w[k]) = l;
        }
    };

    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    // Filter slice of a chunk = 72 planes of 1 KiB in LDS order (8 per tap); wave w streams planes first + w, first + w + 8, ...
    const unsigned lane16 = lane * 16;
    auto dma_filters = [&](int ch, int first, int count, int i0 = 0) {
        const unsigned char* src = bank + ((int64_t)ch * ntile + cotile) * XQ_W_BYTES;
#pragma unroll
        for (int i = i0; i < i0 + count; ++i) {
            const int q = first + wv + 8 * i;
            const unsigned char* g = src + q * 1024;
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(Wl + q * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(g)
                         : "memory");
        }
    };

    // fragment byte offsets of this lane: patch (row 2 wave + row + ky, column 16 half + px + kx, plane = octet), filters (co = 16 i + px)
    const int b_base = oct * XQ_PLANE + ((2 * wave) * XQ_PC + px) * 16;
    const int a_base = oct * 1024 + px * 16;

    f32x4 acc[4][4], master[4][4];  // [16-channel group of the tile][pixel group: row g / 2, column half g % 2]
    {
        const bool with_bias = p.bias != nullptr && p.ksplit <= 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + i * 16 + 4 * oct + r;
                float b0 = 0.f;
                if (with_bias) b0 = p.bias[min(co, p.Cout - 1)];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    master[i][g][r] = b0;
                    acc[i][g][r] = 0.f;
                }
            }
    }

    f16x8 bf[4][2], af[2][2];  // patch fragments [pixel group][part]; filter fragments [buffer][part]
    auto load_bg = [&](int g, int part, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        bf[g][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 4 * XQ_PLANE + (((g >> 1) + ky) * XQ_PC + (g & 1) * 16 + kx) * 16);
    };
    auto load_ai = [&](int buf, int tap, int i) {
#pragma unroll
        for (int part = 0; part < 2; ++part) af[buf][part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 4096 + i * 256);
    };
    // one cout group of one tap: 12 MFMAs (smallest terms first).  The filter fragments of the next step are requested first; during
    // the last group of a tap the patch fragments are refreshed IN PLACE for tap `next_tap` behind their last readers (the wave has
    // 256 registers: no second set).  `extra` is scheduled among the MFMAs (XQ_PIPE: VA vector-ALU instructions per MFMA).
    // FRESH: the group's sums so far are folded into the masters (x inv: un-scaling, a power of two) and its accumulators start again
    // from zero - the first MFMA of each takes a zero C operand.
    auto step = [&](int tap, int i, int next_tap, auto fresh, float inv, auto va, auto&& extra) {
        constexpr bool FRESH = decltype(fresh)::value;
        constexpr int VA = decltype(va)::value;
        const int cur = i & 1;
        const bool refresh = i == 3 && next_tap >= 0;
        XQ_FENCE();
        if (i < 3) load_ai(cur ^ 1, tap, i + 1);
        else if (next_tap >= 0) load_ai(cur ^ 1, next_tap, 0);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FRESH) {
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(acc[i][g]));  // (pinned behind the fence: see split_item)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) master[i][g][r] = fmaf(acc[i][g][r], inv, master[i][g][r]);
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(master[i][g]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][1], bf[g][0], FRESH ? zero : acc[i][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][0], bf[g][0], acc[i][g], 0, 0, 0);
            if (refresh) load_bg(g, 0, next_tap);
        }
        extra();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][0], bf[g][1], acc[i][g], 0, 0, 0);
            if (refresh) load_bg(g, 1, next_tap);
        }
        XQ_PIPE(VA);
        XQ_FENCE();
    };
    auto nothing = []() {};
    constexpr std::true_type FOLD{};
    constexpr std::false_type KEEP{};
    constexpr std::integral_constant<int, 0> V0{};
    constexpr std::integral_constant<int, 2> V2{};
    constexpr std::integral_constant<int, 4> V4{};

    const int nchunks_all = p.Cin / 32;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);
    float inv_prev = 0.f, inv_cur = 0.f, inv_next = 0.f;  // un-scaling factors: previous chunk (its taps 5-8 wait in acc), this one, the next
    if (ch_begin < nchunks) {
        dma_filters(ch_begin, 0, 9);
        load_patch_part(patch_rsrc(ch_begin), 0, 8);
        publish_max();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        inv_cur = chunk_scale() * w_inv_scale;
        inv_next = inv_cur;
#pragma unroll
        for (int k = 0; k < XQ_NI; ++k) split_item(k);
        store_patch();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

#ifdef XQ_STAMP
    if (lane == 0 && p.mask) {
        float* d_ = const_cast<float*>(p.mask) + ((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + 63) * 8;
        d_[0] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4));
        d_[1] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[2] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif
    // chunk c:
    //   tap 0 [fold of the previous chunk's taps 5-8]  taps 0-3 [two channels of patch(c+1) requested per tap; the four planes of
    //   taps 5-8 of THIS chunk a wave streams]   tap 4 [maximum of patch(c+1) -> LDS]   | XM |   scale(c+1),
    //   tap 5 [fold of taps 0-4]  taps 5-8 [the five planes of taps 0-4 (c+1); patch(c+1) split between the MFMAs]
    //   | X1 |   patch(c+1) -> LDS   | X2 |   next chunk
    // (no branch around anything that defines registers: a conditional load makes the compiler wait for it at the join, a conditional
    //  consumer lets it sink the producers into the branch.  The last chunk stages itself once more; only the filter DMA - no
    //  register results - is skipped.)
    for (int ch = ch_begin; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        const bool later = ch > ch_begin;
        const __amdgpu_buffer_rsrc_t rs = patch_rsrc(more ? ch + 1 : ch);
        XQ_MARK(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            load_bg(g, 0, 0);
            load_bg(g, 1, 0);
        }
        load_ai(0, 0, 0);
        step(0, 0, 1, FOLD, inv_prev, V2, [&]() { load_patch_part(rs, 0, 1); });
        step(0, 1, 1, FOLD, inv_prev, V2, nothing);
        step(0, 2, 1, FOLD, inv_prev, V2, [&]() { load_patch_part(rs, 1, 2); });
        step(0, 3, 1, FOLD, inv_prev, V2, [&]() { if (later) dma_filters(ch, 40, 1, 0); });
#pragma unroll
        for (int tap = 1; tap < 4; ++tap) {
            step(tap, 0, tap + 1, KEEP, 0.f, V0, [&]() { load_patch_part(rs, 2 * tap, 2 * tap + 1); });
            step(tap, 1, tap + 1, KEEP, 0.f, V0, nothing);
            step(tap, 2, tap + 1, KEEP, 0.f, V0, [&]() { load_patch_part(rs, 2 * tap + 1, 2 * tap + 2); });
            step(tap, 3, tap + 1, KEEP, 0.f, V0, [&]() { if (later) dma_filters(ch, 40, 1, tap); });
        }
        XQ_MARK(1);
        step(4, 0, 5, KEEP, 0.f, V0, nothing);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step(4, 1, 5, KEEP, 0.f, V4, [&]() { publish_max(); });
        step(4, 2, 5, KEEP, 0.f, V4, nothing);
        step(4, 3, 5, KEEP, 0.f, V0, nothing);   // requests tap 5: the patch and the planes of taps 5-8 stay across XM
        XQ_MARK(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // XM: the filter planes of taps 0-4 are free; the maxima of the next chunk are visible
        XQ_MARK(3);
        inv_next = chunk_scale() * w_inv_scale;
        step(5, 0, 6, FOLD, inv_cur, V2, nothing);
        step(5, 1, 6, FOLD, inv_cur, V2, [&]() { if (more) dma_filters(ch + 1, 0, 1, 0); });
        step(5, 2, 6, FOLD, inv_cur, V2, nothing);
        step(5, 3, 6, FOLD, inv_cur, V2, [&]() { if (more) dma_filters(ch + 1, 0, 1, 1); });
        step(6, 0, 7, KEEP, 0.f, V4, [&]() { split_item(0); });
        step(6, 1, 7, KEEP, 0.f, V4, [&]() { split_item(1); });
        step(6, 2, 7, KEEP, 0.f, V4, [&]() { split_item(2); });
        step(6, 3, 7, KEEP, 0.f, V0, [&]() { if (more) dma_filters(ch + 1, 0, 1, 2); });
        step(7, 0, 8, KEEP, 0.f, V4, [&]() { split_item(3); });
        step(7, 1, 8, KEEP, 0.f, V4, [&]() { split_item(4); });
        step(7, 2, 8, KEEP, 0.f, V0, nothing);
        step(7, 3, 8, KEEP, 0.f, V0, [&]() { if (more) dma_filters(ch + 1, 0, 1, 3); });
        step(8, 0, -1, KEEP, 0.f, V0, nothing);
        step(8, 1, -1, KEEP, 0.f, V0, [&]() { if (more) dma_filters(ch + 1, 0, 1, 4); });
        step(8, 2, -1, KEEP, 0.f, V0, nothing);
        step(8, 3, -1, KEEP, 0.f, V0, nothing);
        XQ_MARK(4);
        inv_prev = inv_cur;
        inv_cur = inv_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X1: every wave is done reading the patch and the remaining filter planes
        XQ_MARK(5);
        store_patch();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X2: patch and the filters of taps 0-4 of the next chunk are in LDS
        XQ_MARK(6);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)  // the last chunk's taps 5-8
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) master[i][g][r] = fmaf(acc[i][g][r], inv_prev, master[i][g][r]);

#ifdef XQ_STAMP
    if (lane == 0 && p.mask) {
        float* d_ = const_cast<float*>(p.mask) + ((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + 63) * 8;
        d_[3] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
        d_[4] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memrealtime());
    }
#endif
    // epilogue: lane holds pixel column 16 (g & 1) + px of row y0 + 2 wr + (g >> 1); register r of group i is output channel 32 wc + 16 i + 4 oct + r
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
    const bool full = co0 + XQ_COT <= p.Cout;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = y0 + 2 * wave + (g >> 1), ox = x0 + (g & 1) * 16 + px;
        const bool pvalid = oy < p.OH && ox < p.OW;
        const int64_t opix = (int64_t)oy * p.OW + ox;
        if (p.ksplit > 1) {  // split-K: un-scaled partial sums, finished by conv_splitk_finish_kernel in split order
            float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * out_plane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + i * 16 + 4 * oct + r;
                    if (pvalid && co < p.Cout) wsp[(int64_t)co * out_plane + opix] = master[i][g][r];
                }
            continue;
        }
        XQ_FENCE();  // (one pixel group at a time: hoisting the next group's loads costs registers the wave does not have)
        if (!pvalid) continue;
        const int cl = co0 + 4 * oct;  // the lane's first output channel
        const int64_t lane_off = (int64_t)cl * out_plane + opix;
        float* __restrict__ yl = yout + lane_off;
        const float* __restrict__ oml = OM ? om + lane_off : nullptr;
        float prev[16], msk[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cr = i * 16 + r;
                const bool cv = full || cl + cr < p.Cout;
                prev[i * 4 + r] = 0.f;
                msk[i * 4 + r] = 1.f;
                if constexpr (ACC) if (cv) prev[i * 4 + r] = yl[(int64_t)cr * out_plane];
                if constexpr (OM) if (cv) msk[i * 4 + r] = oml[(int64_t)cr * out_plane];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cr = i * 16 + r;
                float v = master[i][g][r] + prev[i * 4 + r];
                if (p.relu) v = v > 0.f ? v : 0.f;
                v = msk[i * 4 + r] > 0.f ? v : 0.f;
                if (full || cl + cr < p.Cout) yl[(int64_t)cr * out_plane] = v;
            }
    }
}

// split-K over 32-channel chunks when the grid leaves most of the 256 workgroup slots (1 per CU) empty
static int x3q_choose_split(const ConvArgs& a, int n) {
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + XQ_ROWS - 1) / XQ_ROWS) * ((a.Cout + XQ_COT - 1) / XQ_COT) * split_batch_hint();
    (void)n;  // (the policy looks at the frames the job plans per launch, not at this launch's batch: conv_x3w.hip)
    const int nchunks = a.Cin / 32;
    static const int forced = [] {  // experiments: MAUA_X3Q_KS=k splits every launch k ways (when the layer has the chunks)
        const char* e = getenv("MAUA_X3Q_KS");
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return forced <= nchunks / 2 ? forced : (nchunks >= 4 ? nchunks / 2 : 1);
    if (wgs >= 1024 || nchunks < 4) return 1;  // (four rounds and more: the tail is small)
    const double out_mb = (double)split_batch_hint() * a.Cout * a.OH * a.OW * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 2; ++ks) {
        const double rounds = (double)((wgs * ks + 255) / 256);
        double cost = rounds * ((double)((nchunks + ks - 1) / ks) + 0.7) * 9.0;  // ~9 us per 32-channel chunk of a full CU
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

bool conv_x3q_supports(const ConvArgs& a) {
    return a.Cin % 32 == 0 && (int64_t)a.H * a.W <= (1ll << 24) && a.pad >= 0 && a.pad <= 2;
}

#ifdef XQ_STAMP
static float* g_xq_stamp = nullptr;
extern "C" void maua_xq_set_stamp_buffer(float* buf) { g_xq_stamp = buf; }
#endif

template <bool ACC, bool OM>
static int xq_allow_lds() {  // once per instantiation: the kernel's dynamic LDS is above the default limit
    static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3q_kernel<ACC, OM>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_LDS_BYTES);
    if (rc != hipSuccess) {
        set_error("conv_x3q: hipFuncSetAttribute: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    return MAUA_OK;
}

int conv_x3q_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
#ifdef XQ_STAMP
    p.mask = g_xq_stamp;
#endif
    p.tiles_x = (a.OW + 31) / 32;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + XQ_ROWS - 1) / XQ_ROWS);
    const int ks = a.ws ? x3q_choose_split(a, n) : 1;
    p.ksplit = ks;
    const int64_t cot = (a.Cout + XQ_COT - 1) / XQ_COT, per_xcd = (tiles + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8), (unsigned)cot, (unsigned)(n * ks));
    const bool acc = ks == 1 && a.accumulate != 0, om = ks == 1 && a.omask != nullptr;
    const float w_inv = 1.f / w_scale;
    int rc;
#define XQ_LAUNCH(...)                                                                             \
    do {                                                                                           \
        rc = xq_allow_lds<__VA_ARGS__>();                                                          \
        if (rc) return rc;                                                                         \
        hipLaunchKernelGGL((conv_x3q_kernel<__VA_ARGS__>), grid, dim3(XQ_THREADS), XQ_LDS_BYTES, stream, p, w_inv); \
    } while (0)
    if (acc && om) XQ_LAUNCH(true, true);
    else if (acc) XQ_LAUNCH(true, false);
    else if (om) XQ_LAUNCH(false, true);
    else XQ_LAUNCH(false, false);
#undef XQ_LAUNCH
    rc = check_launch("conv_x3q_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_x3q_bank_bytes(int cout_produced, int cin_consumed) {
    if (cout_produced <= 0 || cin_consumed <= 0 || cout_produced > (1 << 20) || cin_consumed > (1 << 20)) return 0;
    const size_t nchunk = (cin_consumed + 31) / 32, ntile = (cout_produced + XQ_COT - 1) / XQ_COT;
    return nchunk * ntile * XQ_W_BYTES;
}

int maua_conv_pack_filters_x3q(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                               maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20) && w_scale > 0.f, MAUA_E_INVAL,
                 "conv_pack_filters_x3q: bad args");
    if (bank_fwd) {
        hipLaunchKernelGGL(pack_x3q_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_fwd, cout,
                           cin, 0, w_scale);
        int rc = check_launch("pack_x3q_kernel");
        if (rc) return rc;
    }
    if (bank_bwd) {
        hipLaunchKernelGGL(pack_x3q_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_bwd, cout,
                           cin, 1, w_scale);
        return check_launch("pack_x3q_kernel");
    }
    return MAUA_OK;
}

int maua_conv_x3q_supported(int cin, int h, int w, int pad) {
    ConvArgs a{};
    a.Cin = cin;
    a.H = h;
    a.W = w;
    a.pad = pad;
    return conv_dims_ok(1, cin, h, w, 1, pad) && conv_x3q_supports(a) ? 1 : 0;
}

size_t maua_conv_x3q_workspace_bytes(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const int ks = x3q_choose_split(a, n);
    return ks > 1 ? (size_t)n * ks * cout * a.OH * a.OW * sizeof(float) : 0;
}

int maua_conv_x3q_split(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    return x3q_choose_split(a, n);
}

int maua_conv3x3_x3q(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3q: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3q: bad dims");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_x3q: input smaller than the filter");
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
    MAUA_REQUIRE(conv_x3q_supports(a), MAUA_E_UNSUPPORTED, "conv3x3_x3q: needs cin %% 32 == 0 and a plane of at most 2^24 pixels");
    a.ws = (workspace && workspace_bytes >= maua_conv_x3q_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    return conv_x3q_launch(a, n, w_scale, (hipStream_t)stream);
}

}  // extern "C"
