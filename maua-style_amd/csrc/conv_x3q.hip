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
// hipcc-flags: -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace maua {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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

// The kernel's argument segment (ConvArgs is the first parameter): the in-launch finish re-reads its arguments from here behind the K loop
__device__ __forceinline__ const ConvArgs* xq_kernel_arguments() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}

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

// POOL: the epilogue applies ReLU and the 2x2 / 2 max pool that follows (conv_x3w.hip): `y` is the pooled map, p.pool_codes its decision
// bytes.  UNPOOL: `x` is the POOLED map of a 2x2 / 2 max pool and p.in_codes its decision bytes; the input the convolution sees is the
// pool's backward pass over them (pool2x2_bwd_codes_kernel's arithmetic), rebuilt while staging - the full-size gradient never exists.
template <bool ACC, bool OM, bool POOL = false, bool UNPOOL = false>
__global__ void __launch_bounds__(XQ_THREADS, 2) conv_x3q_kernel(ConvArgs p, float w_inv_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Pl = smem;                    // [part][octet][pos][16 B]
    unsigned char* Wl = smem + XQ_PATCH_BYTES;   // [tap][part][octet][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + XQ_PATCH_BYTES + XQ_W_BYTES);  // per-wave maxima of the chunk being staged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, oct = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#ifdef XQ_STAMP
    const unsigned t_entry_ = (unsigned)__builtin_amdgcn_s_memtime(), r_entry_ = (unsigned)__builtin_amdgcn_s_memrealtime();
#define XQ_EXIT_STAMP()                                                                                                              \
    do {                                                                                                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                             \
        if (lane == 0 && p.mask) {                                                                                                   \
            float* d_ = const_cast<float*>(p.mask) + ((((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64 + 63) * 8;    \
            d_[5] = __builtin_bit_cast(float, t_entry_);                                                                             \
            d_[6] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());                                               \
            d_[7] = __builtin_bit_cast(float, r_entry_);                                                                             \
        }                                                                                                                            \
    } while (0)
#else
#define XQ_EXIT_STAMP() do {} while (0)
#endif
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    // XCD-aware order (conv_x6.hip): XCD k owns the k-th contiguous band of pixel tiles.  Workgroups go to the XCDs in turn in dispatch order
    // (x fastest, gridDim.x % 8 == 0), so `slot` counts an XCD's workgroups in the order they start: with p.cot_inner the channel tiles of a
    // pixel tile take consecutive slots - they run side by side on one XCD and the patch leaves memory once, not once per channel tile.
    const int ntile = gridDim.y;
    const int tiles_total = p.tiles_x * ((p.OH + XQ_ROWS - 1) / XQ_ROWS);
    const int per_xcd = (tiles_total + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = (int)((blockIdx.y * gridDim.x + blockIdx.x) >> 3);
    const int cotile = p.cot_inner ? slot % ntile : (int)blockIdx.y;
    const int tile = xcd * per_xcd + (p.cot_inner ? slot / ntile : (int)(blockIdx.x >> 3));
    const int co0 = cotile * XQ_COT;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const int st_w = UNPOOL ? p.W >> 1 : p.W;                              // row pitch and plane of the array the patch is staged from
    const int st_plane = UNPOOL ? (p.H >> 1) * (p.W >> 1) : in_plane;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * st_plane;
    const unsigned char* __restrict__ xcodes = UNPOOL ? p.in_codes + (int64_t)n * p.Cin * st_plane : nullptr;
    const unsigned code_mask = (unsigned)p.in_code_mask;
    if (tile >= min((xcd + 1) * per_xcd, tiles_total)) return;  // whole workgroup leaves
    const int x0 = (tile % p.tiles_x) * 32, y0 = (tile / p.tiles_x) * XQ_ROWS;
    if (tid == 0) {  // for the in-launch finish of a split channel loop (behind the K loop; read there behind several barriers)
        reinterpret_cast<int*>(Ml)[9] = (n * ntile + cotile) * tiles_total + tile;
        reinterpret_cast<int*>(Ml)[10] = split;
    }

    // Staging items of this thread: item k = (octet, position) number tid + 512 k.  voff = byte offset of the item's first
    // channel from the chunk's first plane; out-of-image positions and items past the end get an offset beyond the buffer's
    // range, for which a buffer load returns 0 (no selects on the values).
    //
    // UNPOOL: two items per thread, and they are POOLED elements - (octet, pooled row, pooled column) number tid + 512 k of the 4 x 10 x 18
    // pooled elements whose 2x2 windows cover the 18 x 34 patch.  The thread loads an element's eight channels and their eight decision
    // bytes (one 8-byte load from the [octet][pooled pixel][8] layout), splits the eight values once, and writes each of the window's
    // four corners that lies inside the patch: channel c's halves where the byte names that corner, zero elsewhere - the patch in LDS
    // is bit for bit what staging pool2x2_bwd_codes_kernel's output would have left.
    constexpr int NI = UNPOOL ? 2 : XQ_NI;
    unsigned voff[NI], lds_w[NI];
    unsigned vcode[NI], inmask[NI];  // UNPOOL: offset of the item's decision bytes; bit q = corner q (2 dy + dx) is a patch position
    if constexpr (UNPOOL) {
        constexpr int PR = XQ_ROWS / 2 + 2, PCW = 18;  // pooled rows / columns under the patch (any pad in 0 ... 2)
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int idx = tid + XQ_THREADS * k;
            const int o = idx / (PR * PCW);
            const int rem = idx - o * PR * PCW;
            const int pr = rem / PCW, pc = rem - pr * PCW;
            const int ppy = ((y0 - p.pad) >> 1) + pr, ppx = ((x0 - p.pad) >> 1) + pc;  // (arithmetic shifts: floor for the -1 / -2 of the first tiles)
            const bool item = idx < 4 * PR * PCW;
            const bool ok = item && ppy >= 0 && ppy < (p.H >> 1) && ppx >= 0 && ppx < st_w;
            const int pidx = ppy * st_w + ppx;
            voff[k] = ok ? (unsigned)(o * 8 * st_plane + pidx) * 4u : 0x80000000u;
            vcode[k] = ok ? (unsigned)(o * st_plane + pidx) * 8u : 0x80000000u;
            const int r0 = 2 * ppy - (y0 - p.pad), c0 = 2 * ppx - (x0 - p.pad);  // patch position of the window's first corner (-1 ... )
            inmask[k] = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + (q >> 1), c = c0 + (q & 1);
                if (item && r >= 0 && r < XQ_PR && c >= 0 && c < XQ_PC) inmask[k] |= 1u << q;
            }
            lds_w[k] = (unsigned)(o * XQ_PLANE + (r0 * XQ_PC + c0) * 16);  // (of corner 0; only ever used plus a corner's offset, for corners inside)
        }
    } else {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int idx = tid + XQ_THREADS * k;
            const int o = idx / XQ_NPOS;
            const int pos = idx - o * XQ_NPOS;
            const int r = pos / XQ_PC, c = pos - r * XQ_PC;
            const int iy = y0 + r - p.pad, ix = x0 + c - p.pad;
            const bool ok = idx < XQ_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            voff[k] = ok ? (unsigned)(o * 8 * in_plane + iy * p.W + ix) * 4u : 0x80000000u;
            lds_w[k] = idx < XQ_ITEMS ? (unsigned)(o * XQ_PLANE + pos * 16) : (unsigned)(XQ_NPOS_PAD - 1) * 16u;  // (items past the end: a padding slot)
            vcode[k] = inmask[k] = 0;
        }
    }
    const unsigned range = (unsigned)st_plane * 128u;  // 32 planes from the chunk's first one: the range check sees the vector offset only
    float rp[NI][8];       // the next chunk's patch: raw values, then (in place) the packed fp16 halves [0..3] high, [4..7] low
    u32x2 cd[NI];          // UNPOOL: the items' decision bytes
    unsigned sel2[NI][4];  // UNPOOL: the decisions two per register (16-bit lanes, channels 2 i and 2 i + 1), ReLU bit masked as asked
    struct PatchSrc {
        __amdgpu_buffer_rsrc_t x, codes;
    };
    auto patch_rsrc = [&](int ch) {
        asm volatile("" : "+s"(ch));
        PatchSrc r;
        r.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (int64_t)ch * 32 * st_plane), 0, range, 0x00020000);
        r.codes = r.x;
        if constexpr (UNPOOL)
            r.codes = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xcodes + (int64_t)ch * 32 * st_plane), 0, range >> 2, 0x00020000);
        return r;
    };
    // channels [c_lo, c_hi) of every item (UNPOOL: the decision bytes come with channel 0)
    auto load_patch_part = [&](const PatchSrc& rs, int c_lo, int c_hi) {
#ifdef XQ_NO_SPLIT
        // TIMING-ONLY diagnostic (tools/nosplit_ceiling.sh; wrong numbers): what the K loop would cost if the patch came
        // as ready fp16 pairs by LDS-DMA - ten 1 KiB buffer-to-LDS loads per wave and chunk (the patch's 78 KiB) instead of forty dword
        // loads, the maximum, the split and ten ds_write_b128 per thread.
#if defined(__HIP_DEVICE_COMPILE__)
        for (int c = c_lo; c < c_hi; ++c)
            if (c < 5)
                for (int part = 0; part < 2; ++part)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.x, (__attribute__((address_space(3))) void*)(Pl + ((wv * 5 + c) * 2 + part) * 992), 16,
                                                             voff[c < NI ? c : 0], part * 64, 0, 0);
#endif
        return;
#endif
        if constexpr (UNPOOL) {
            if (c_lo == 0) {
#pragma unroll
                for (int k = 0; k < NI; ++k) cd[k] = __builtin_amdgcn_raw_buffer_load_b64(rs.codes, vcode[k], 0, 0);
            }
        }
#pragma unroll
        for (int c = c_lo; c < c_hi; ++c)
#pragma unroll
            for (int k = 0; k < NI; ++k)
                rp[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs.x, voff[k], c * st_plane * 4, 0));
    };
    auto publish_max = [&]() {
#ifdef XQ_NO_SPLIT
        if (lane == 0) Ml[wave] = 1.f;
        return;
#endif
        if constexpr (UNPOOL) {
            // A value counts (for the chunk's scale, and at all) where its byte names a corner that is a patch position - bit 2 of the
            // byte, kept by code_mask = 7, names none.  What another tile's patch holds of this window is that tile's business.
            const unsigned m2 = code_mask * 0x00010001u;
#pragma unroll
            for (int k = 0; k < NI; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sel2[k][i] = __builtin_amdgcn_perm(0u, cd[k][i >> 1], (i & 1) ? 0x0c030c02u : 0x0c010c00u) & m2;
                    const unsigned s0 = sel2[k][i] & 0xffffu, s1 = sel2[k][i] >> 16;
                    rp[k][2 * i] = ((inmask[k] >> s0) & 1u) ? rp[k][2 * i] : 0.f;
                    rp[k][2 * i + 1] = ((inmask[k] >> s1) & 1u) ? rp[k][2 * i + 1] : 0.f;
                }
        }
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < NI; ++k)
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
#ifdef XQ_NO_SPLIT
        return;
#endif
        if (k >= NI) return;
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
#ifdef XQ_NO_SPLIT
        return;
#endif
        if constexpr (UNPOOL) {
#pragma unroll
            for (int k = 0; k < NI; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((inmask[k] >> q) & 1u) {
                        const unsigned dst = lds_w[k] + (unsigned)(((q >> 1) * XQ_PC + (q & 1)) * 16);
                        u32x4 h, l;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {  // dword i of corner q's vector: the halves of the channels whose byte names q
                            const unsigned t = sel2[k][i] ^ ((unsigned)q * 0x00010001u);                        // 16-bit lane == 0 where it does
                            const unsigned keep = ((t & 0xffffu) ? 0u : 0xffffu) | ((t >> 16) ? 0u : 0xffff0000u);
                            h[i] = __builtin_bit_cast(unsigned, rp[k][i]) & keep;
                            l[i] = __builtin_bit_cast(unsigned, rp[k][4 + i]) & keep;
                        }
                        *reinterpret_cast<u32x4*>(Pl + dst) = h;
                        *reinterpret_cast<u32x4*>(Pl + 4 * XQ_PLANE + dst) = l;
                    }
            return;
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
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
        else if (next_tap >= 0 && tap != 4) load_ai(cur ^ 1, next_tap, 0);  // (tap 5's planes are in LDS behind XM only: see the loop)
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
        for (int k = 0; k < NI; ++k) split_item(k);
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
        d_[-8] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20));  // slot 62: HW_REG_XCC_ID
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
        const PatchSrc rs = patch_rsrc(more ? ch + 1 : ch);
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
        step(4, 3, 5, KEEP, 0.f, V0, nothing);   // requests tap 5's PATCH fragments (the patch stays across XM) - not its filter fragments:
                                                 // the planes of taps 5-8 were streamed during taps 0-3 by all eight waves, and another
                                                 // wave's LDS-DMA is only known to have landed behind the barrier that follows its wait
        XQ_MARK(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // XM: the filter planes of taps 0-4 are free, those of taps 5-8 and the maxima of the next chunk visible
        XQ_MARK(3);
        load_ai(0, 5, 0);
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
    // Split channel loop finished INSIDE the launch (round 6; p.arrive = one zeroed arrival counter per (image, channel tile, pixel tile)):
    // every split's workgroup leaves its un-scaled partial sums in its slab - lane-linear, sixteen 16-byte vectors per thread, written
    // THROUGH (sc1), drained - and draws a ticket (one relaxed agent-scope add per workgroup).  The workgroup that draws the last ticket of
    // its tile adds the slabs in split order - its own sums from its registers at their place in that order, the others by sc1 loads: the
    // additions of conv_splitk_finish_kernel, bit for bit - then the bias, and runs the one-pass epilogue below (mask, ReLU, pool,
    // accumulation); the others leave.  Nothing waits for anything: whatever the order in which the workgroups of a tile run, the last one
    // to arrive finds every other slab complete (cdna_hip_programming.md, in-launch split-K reduction, sc1 form: the stores are drained in
    // front of the workgroup's barrier, the ticket add follows the barrier, every load of a slab is an sc1 load).  The last arriver puts
    // the counter back to zero: the next launch (stream order) finds it so.
    // (The K loop has no scalar register to spare - one more value alive across it and the allocator spills staging addresses into the
    //  loop.  So this phase takes nothing across the loop: its tile number and split index were parked in LDS by thread 0 at entry, and
    //  the kernel arguments it needs are read again from the argument segment through a pointer the compiler cannot see through.)
    bool final_sums = p.ksplit <= 1;
    const ConvArgs* kp = xq_kernel_arguments();
    asm volatile("" : "+s"(kp));
    if (kp->ksplit > 1 && kp->arrive != nullptr) {
        const int ksplit = kp->ksplit;
        const int unit = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(Ml)[9]);
        const int split = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(Ml)[10]);
        unsigned* const arrive = kp->arrive + unit;
        const float* const bias = kp->bias;
        constexpr unsigned SLAB = XQ_COT * XQ_ROWS * 32 * 4;  // 128 KiB
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(kp->ws) + (int64_t)unit * ksplit * SLAB, 0,
                                                                            (unsigned)ksplit * SLAB, 0x00020000);
        const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, master[q >> 2][q & 3]), srs, toff, (unsigned)split * SLAB + q * 8192u, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned* tick = reinterpret_cast<unsigned*>(Ml) + 8;
        if (tid == 0) *tick = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if ((int)*tick != ksplit - 1) {
            XQ_EXIT_STAMP();
            return;
        }
        if (tid == 0) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ksplit; ++k) {
            if (k == split) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[i][g] += master[i][g];
            } else {
#ifndef XQ_FIN_BATCH
#define XQ_FIN_BATCH 8
#endif
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += XQ_FIN_BATCH) {
                    u32x4 t[XQ_FIN_BATCH];
#pragma unroll
                    for (int q = 0; q < XQ_FIN_BATCH; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(srs, toff, (unsigned)k * SLAB + (q0 + q) * 8192u, 16);
#pragma unroll
                    for (int q = 0; q < XQ_FIN_BATCH; ++q) acc[(q0 + q) >> 2][(q0 + q) & 3] += __builtin_bit_cast(f32x4, t[q]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + i * 16 + 4 * oct + r;
                const float b0 = bias != nullptr ? bias[min(co, kp->Cout - 1)] : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) master[i][g][r] = acc[i][g][r] + b0;
            }
        final_sums = true;
    }
    // epilogue: lane holds pixel column 16 (g & 1) + px of row y0 + 2 wave + (g >> 1); register r of group i is output channel 16 i + 4 oct + r
    if constexpr (POOL) if (final_sums) {
        // ReLU + the 2x2 / 2 max pool behind it: a wave's two rows and neighbouring lanes are exactly the windows, so the full-size
        // activation never goes to memory - only the pooled map and one decision byte per window (what pool2x2_fwd_codes_kernel
        // leaves: position of the first maximum in scan order, bit 2 = the maximum is <= 0; bytes laid out [octet of channels][pooled
        // pixel][8 channels], Cout % 8 == 0): the lane's four consecutive channels of a group are one dword of that layout.
        const int PW = p.OW >> 1;
        const int64_t pplane = (int64_t)(p.OH >> 1) * PW;
        float* __restrict__ py = p.y + (int64_t)n * p.Cout * pplane;
        unsigned char* __restrict__ pc = p.pool_codes + (int64_t)n * p.Cout * pplane;
        const int oy = y0 + 2 * wave;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // column half of the tile
            const int oxx = x0 + 16 * h + px;
            const bool store = (lane & 1) == 0 && oy + 1 < p.OH && oxx + 1 < p.OW;  // (a window is inside or outside; an odd plane's last row / column has none)
            const int64_t ppix = (int64_t)(oy >> 1) * PW + (oxx >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cq = co0 + 16 * i + 4 * oct;
                unsigned pk = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = master[i][h][r], c = master[i][2 + h][r];
                    a = a > 0.f ? a : 0.f;
                    c = c > 0.f ? c : 0.f;
                    const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x101, 0xf, 0xf, false));  // row_shl:1
                    const float d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x101, 0xf, 0xf, false));
                    float m = a;
                    unsigned arg = 0;
                    if (b > m) { m = b; arg = 1; }
                    if (c > m) { m = c; arg = 2; }
                    if (d > m) { m = d; arg = 3; }
                    pk |= (arg | (m > 0.f ? 0u : 4u)) << (8 * r);
                    if (store && cq < p.Cout) py[(int64_t)(cq + r) * pplane + ppix] = m;
                }
                if (store && cq < p.Cout) *reinterpret_cast<unsigned*>(pc + ((int64_t)(cq >> 3) * pplane + ppix) * 8 + (cq & 7)) = pk;
            }
        }
        XQ_EXIT_STAMP();
        return;
    }
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
    const bool full = co0 + XQ_COT <= p.Cout;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = y0 + 2 * wave + (g >> 1), ox = x0 + (g & 1) * 16 + px;
        const bool pvalid = oy < p.OH && ox < p.OW;
        const int64_t opix = (int64_t)oy * p.OW + ox;
        if (!final_sums) {  // split-K: un-scaled partial sums, finished by conv_splitk_finish_kernel in split order
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
        int64_t lane_off = (int64_t)cl * out_plane + opix;
        // (an address the compiler cannot form before this point: the previous contents and the mask are loop-invariant loads of
        //  `restrict` arrays, and hoisted above the K loop they would cost 32 registers the wave does not have)
        if constexpr (ACC || OM) asm volatile("" : "+v"(lane_off));
        float* __restrict__ yl = yout + lane_off;
        const float* __restrict__ oml = OM ? om + lane_off : nullptr;
        // (accumulating launches - rare - take one channel group at a time: with the previous contents of all sixteen channels in flight
        //  the compiler's allocation of the whole kernel tips over into hundreds of spills; masks alone come sixteen at a time)
        constexpr int GB = ACC ? 1 : 4;  // channel groups per batch of loads
#pragma unroll
        for (int i0 = 0; i0 < 4; i0 += GB) {
            if constexpr (ACC) XQ_FENCE();
            float prev[GB * 4], msk[GB * 4];
#pragma unroll
            for (int i = i0; i < i0 + GB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cr = i * 16 + r, e = (i - i0) * 4 + r;
                    const bool cv = full || cl + cr < p.Cout;
                    prev[e] = 0.f;
                    msk[e] = 1.f;
                    if constexpr (ACC) if (cv) prev[e] = yl[(int64_t)cr * out_plane];
                    if constexpr (OM) if (cv) msk[e] = oml[(int64_t)cr * out_plane];
                }
#pragma unroll
            for (int i = i0; i < i0 + GB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cr = i * 16 + r, e = (i - i0) * 4 + r;
                    float v = master[i][g][r] + prev[e];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    v = msk[e] > 0.f ? v : 0.f;
                    if (full || cl + cr < p.Cout) yl[(int64_t)cr * out_plane] = v;
                }
        }
    }
    XQ_EXIT_STAMP();
}

// split-K over 32-channel chunks when the grid leaves most of the 256 workgroup slots (1 per CU) empty
static int x3q_choose_split(const ConvArgs& a, int n) {
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + XQ_ROWS - 1) / XQ_ROWS) * ((a.Cout + XQ_COT - 1) / XQ_COT) * split_batch_hint();
    (void)n;  // (the policy looks at the frames the job plans per launch, not at this launch's batch: conv_x3w.hip)
    const int nchunks = a.Cin / 32;
    const int forced = (int)tuning("x3q_ks", 0);  // experiments: splits every launch k ways (when the layer has the chunks)
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

template <bool ACC, bool OM, bool POOL, bool UNPOOL>
static int xq_launch_one(const ConvArgs& p, dim3 grid, float w_inv, hipStream_t stream) {
    // once per instantiation and device: the kernel's 150 KiB of dynamic LDS are above the default limit
    static unsigned long long served = 0;
    const hipError_t rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&conv_x3q_kernel<ACC, OM, POOL, UNPOOL>), XQ_LDS_BYTES, &served);
    if (rc != hipSuccess) {
        set_error("conv_x3q: hipFuncSetAttribute: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    hipLaunchKernelGGL((conv_x3q_kernel<ACC, OM, POOL, UNPOOL>), grid, dim3(XQ_THREADS), XQ_LDS_BYTES, stream, p, w_inv);
    return check_launch("conv_x3q_kernel");
}

// Bytes of slabs a launch that finishes its split inside the launch needs: one 128 KiB slab per (image, channel tile, pixel tile, split) - whole
// tiles, so a little more than the [split][Cout][OH][OW] slabs of the two-launch form on ragged planes.
static size_t x3q_in_launch_bytes(const ConvArgs& a, int n, int ks) {
    const int64_t tiles = (int64_t)((a.OW + 31) / 32) * ((a.OH + XQ_ROWS - 1) / XQ_ROWS), cot = (a.Cout + XQ_COT - 1) / XQ_COT;
    return (size_t)n * cot * tiles * ks * (XQ_COT * XQ_ROWS * 32 * 4);
}
// Whether a launch split `ks` ways finishes inside the launch: the caller armed this workspace (a.arrive), the split is small enough for
// one workgroup to add the other slabs (tuning constant finish_in_launch_max_ks, default 4: it reads (ks - 1) x 128 KiB alone), and the
// tiles have counters.
static bool x3q_finish_in_launch(const ConvArgs& a, int n, int ks) {
    const int64_t units = (int64_t)n * ((a.Cout + XQ_COT - 1) / XQ_COT) * ((a.OW + 31) / 32) * ((a.OH + XQ_ROWS - 1) / XQ_ROWS);
    return ks > 1 && a.arrive != nullptr && ks <= (int)tuning("finish_in_launch_max_ks", 4) && units <= ARRIVE_COUNTERS;
}

int conv_x3q_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
#ifdef XQ_STAMP
    p.mask = g_xq_stamp;
#endif
    p.tiles_x = (a.OW + 31) / 32;
    p.cot_inner = tuning("cot_inner", 0) != 0 ? 1 : 0;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + XQ_ROWS - 1) / XQ_ROWS);
    const int ks = a.ws ? x3q_choose_split(a, n) : 1;
    p.ksplit = ks;
    const int64_t cot = (a.Cout + XQ_COT - 1) / XQ_COT, per_xcd = (tiles + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8), (unsigned)cot, (unsigned)(n * ks));
    const bool whole = ks == 1 || x3q_finish_in_launch(a, n, ks);  // the launch's epilogue holds complete sums (one pass, or the last arriver's)
    if (!whole) p.arrive = nullptr;
    const bool acc = whole && a.accumulate != 0, om = whole && a.omask != nullptr;
    const float w_inv = 1.f / w_scale;
    int rc;
    if (a.pool_codes && whole) rc = xq_launch_one<false, false, true, false>(p, grid, w_inv, stream);
    else if (a.in_codes && om) rc = xq_launch_one<false, true, false, true>(p, grid, w_inv, stream);
    else if (a.in_codes) rc = xq_launch_one<false, false, false, true>(p, grid, w_inv, stream);
    else if (acc && om) rc = xq_launch_one<true, true, false, false>(p, grid, w_inv, stream);
    else if (acc) rc = xq_launch_one<true, false, false, false>(p, grid, w_inv, stream);
    else if (om) rc = xq_launch_one<false, true, false, false>(p, grid, w_inv, stream);
    else rc = xq_launch_one<false, false, false, false>(p, grid, w_inv, stream);
    if (rc || whole) return rc;
    // (a split channel loop leaves partial sums: the ReLU + pool of a pooling launch then happen in the pass that adds them)
    return a.pool_codes ? conv_splitk_finish_pool(a, n, ks, stream) : conv_splitk_finish(a, n, ks, stream);
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
    if (ks <= 1) return 0;
    const size_t two_launches = (size_t)n * ks * cout * a.OH * a.OW * sizeof(float);
    const size_t in_launch = x3q_in_launch_bytes(a, n, ks);  // (whole tiles; a caller that arms its workspace gets this form)
    return in_launch > two_launches ? in_launch : two_launches;
}

int maua_conv_x3q_preferred(int n, int cin, int h, int w, int cout, int pad) {
    // One fat workgroup per CU only pays where the grid fills the 256 slots evenly and every workgroup has a K loop long enough to
    // carry its exposed prologue and epilogue (measured, profiles/probes_r04.md section 2: at 724 px - 181 x 181 and 90 x 90 planes,
    // 288 and 432 workgroups - conv_x3w's finer tiles win by 15-20 %; at 1024 / 2048 px the grids are multiples of 256).
    if (!conv_dims_ok(n, cin, h, w, cout, pad) || cin % 32 != 0) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const double min_fill = tuning("x3q_min_fill", 0.85);
    const int min_chunks = (int)tuning("x3q_min_chunks", 4);
    const int ks = x3q_choose_split(a, n);
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + XQ_ROWS - 1) / XQ_ROWS) * ((a.Cout + XQ_COT - 1) / XQ_COT) * split_batch_hint() * ks;
    const double fill = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    // (ragged tiles count too: the pixels a 16 x 32 tile grid covers beyond the plane)
    const double cover = (double)a.OH * a.OW / ((double)((a.OH + XQ_ROWS - 1) / XQ_ROWS * XQ_ROWS) * ((a.OW + 31) / 32 * 32));
    return fill * cover >= min_fill && (cin / 32 + ks - 1) / ks >= min_chunks ? 1 : 0;
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

static int conv3x3_x3q_entry(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y, int n,
                             int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace, size_t workspace_bytes,
                             maua_stream_t stream, const unsigned char* in_codes = nullptr, int in_code_mask = 0) {
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
    if (in_codes) {
        MAUA_REQUIRE(h >= 2 && w >= 2 && !accumulate, MAUA_E_UNSUPPORTED, "conv3x3_x3q_unpool: needs an input plane of 2 x 2 and more, no accumulation");
        a.in_codes = in_codes;
        a.in_code_mask = in_code_mask;
    }
    a.ws = (workspace && workspace_bytes >= maua_conv_x3q_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    a.arrive = a.ws ? armed_counters(workspace) : nullptr;  // (the calling thread armed this workspace: small splits finish inside the launch)
    return conv_x3q_launch(a, n, w_scale, (hipStream_t)stream);
}

int maua_conv3x3_x3q(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream) {
    return conv3x3_x3q_entry(x, bank, w_scale, bias, out_relu_mask, y, n, cin, h, w, cout, pad, relu, accumulate, workspace, workspace_bytes,
                             stream);
}

int maua_conv3x3_x3q_relu_pool(const float* x, const void* bank, float w_scale, const float* bias, float* pooled, unsigned char* codes,
                               int n, int cin, int h, int w, int cout, int pad, void* workspace, size_t workspace_bytes,
                               maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && pooled && codes && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3q_relu_pool: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3q_relu_pool: bad dims");
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
    MAUA_REQUIRE(a.OH >= 2 && a.OW >= 2 && cout % 8 == 0 && conv_x3q_supports(a), MAUA_E_UNSUPPORTED,
                 "conv3x3_x3q_relu_pool: needs an output plane of 2 x 2 and more, cin %% 32 == 0, cout %% 8 == 0");
    // without a workspace: one pass over the channels, the epilogue holds complete sums and pools them itself
    a.ws = (workspace && workspace_bytes >= maua_conv_x3q_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
    a.arrive = a.ws ? armed_counters(workspace) : nullptr;  // (the calling thread armed this workspace: small splits finish inside the launch)
    return conv_x3q_launch(a, n, w_scale, (hipStream_t)stream);
}

int maua_conv3x3_x3q_unpool(const float* pooled_x, const unsigned char* codes, int honour_relu_bit, const void* bank, float w_scale,
                            const float* out_relu_mask, float* y, int n, int cin, int h, int w, int cout, int pad, void* workspace,
                            size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(codes, MAUA_E_INVAL, "conv3x3_x3q_unpool: null decision bytes");
    return conv3x3_x3q_entry(pooled_x, bank, w_scale, nullptr, out_relu_mask, y, n, cin, h, w, cout, pad, 0, 0, workspace, workspace_bytes, stream,
                             codes, honour_relu_bit ? 7 : 3);
}

}  // extern "C"
