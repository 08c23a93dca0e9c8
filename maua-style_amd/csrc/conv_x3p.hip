// 3x3 stride-1 convolution in fp16x3 arithmetic, fourth structure: conv_x3q.hip's workgroup made PERSISTENT.
//
// conv_x3q.hip runs one workgroup of eight waves per CU (150 KiB of LDS); nothing runs beside a workgroup's prologue (first patch
// + nine filter taps: 150 KB in one burst while every other CU does the same) and its epilogue (64 values per lane; with a ReLU
// mask four dependent load -> store rounds).  In-kernel stamps (tools/x3q_phases.py, profiles/probes_r05.md): on conv1_2's shape
// (64 -> 64 @ 1024 x 1024, two 32-channel chunks per tile) a wave spends 21-25 k cycles in the prologue, 42 k in the K loop and
// 10 k (plain) / 20 k (pool) / 38 k (masked) in the epilogue - the matrix pipe idles for half of its life; at four chunks a quarter,
// at sixteen 7 %.  This kernel walks a static list of work items per workgroup - item = (image, output-channel tile, pixel tile,
// K split) - as ONE stream of 32-channel chunks:
//   * the first chunk of the next item is staged during the last chunk of the current one exactly like any next chunk (the
//     staging offsets are recomputed for the new tile at the top of that chunk): no prologue after the first;
//   * an item's epilogue rides in the first chunk of the NEXT item: group i of the masters is final once tap 0's fold of group i
//     has run, and free again when tap 5 folds into it - twenty steps in which its sixteen registers are un-scaled, masked /
//     pooled, stored (buffer stores: tile base and channel in the scalar offset, the lane's pixel in one vector register, image
//     borders as out-of-range offsets) and reset to the bias (kept in LDS);
//   * the Gram backward of a style loss on the layer's output-side map (conv_x3w.hip's fused form, D . F as one-tap chunks) runs
//     between an item's last chunk and the next item's first, from registers only.
// Arithmetic, LDS layout, chunk pipeline and filter banks are conv_x3q.hip's: a one-pass launch gives the same bits.
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

constexpr int XP_COT = 64;
constexpr int XP_ROWS = 16, XP_PR = 18, XP_PC = 34;
constexpr int XP_NPOS = XP_PR * XP_PC;           // 612
constexpr int XP_NPOS_PAD = 624;                 // a multiple of 16 positions
constexpr int XP_PLANE = XP_NPOS_PAD * 16;       // bytes of one [pos][8 ch] plane
constexpr int XP_PATCH_BYTES = 8 * XP_PLANE;     // [part][octet]
constexpr int XP_TAP_BYTES = 2 * 4 * XP_COT * 16;  // [part][octet][co][16 B] = 8 planes of 1 KiB
constexpr int XP_W_BYTES = 9 * XP_TAP_BYTES;
constexpr int XP_NI = 5;
constexpr int XP_THREADS = 512;
constexpr int XP_BIAS_MAX = 512;                 // output channels whose bias fits the LDS table
constexpr int XP_LDS_BYTES = XP_PATCH_BYTES + XP_W_BYTES + 64 + XP_BIAS_MAX * 4 + XP_THREADS * 12;  // + three words of staging geometry per thread
constexpr unsigned XP_OOB = 0x80000000u;
// Schedule constants (tools/build_x3p_variant.sh builds libraries with others for A/B runs on one box):
#ifndef XP_W1_PIECES
#define XP_W1_PIECES 4     // epilogue pieces (of 16) that ride in taps 0-4 of an item's first chunk; the others ride in taps 5-7
#endif
#ifndef XP_EARLY_LOADS
#define XP_EARLY_LOADS 0   // 1: every chunk requests the next patch in steps 0-7 (0: the chunks without an epilogue spread it over taps 0-3)
#endif
#ifndef XP_MIDFOLD
#define XP_MIDFOLD 1       // chunks that are not the first of an item fold their sums into the masters at tap 5 as well (0: one fold per chunk everywhere)
#endif
#ifndef XP_STAGGER
#define XP_STAGGER 0       // start delay of workgroup q: (q % 4) x this many 1024-cycle sleeps
#endif

// step (4 tap + group, 0 ... 35) of an item's first chunk in which piece k of the previous item's epilogue rides
__host__ __device__ constexpr int xp_first_pieces(bool pool) { return pool ? XP_W1_PIECES / 2 : XP_W1_PIECES; }
__host__ __device__ constexpr int xp_step_of(bool pool, int k) {
    const int p1 = xp_first_pieces(pool);
    return pool ? (k < p1 ? 16 - 2 * (p1 - 1 - k) : 20 + 2 * (k - p1)) : (k < p1 ? 17 - p1 + k : 20 + (k - p1));
}

struct X3pArgs {
    int items;    // work items of the launch: ((n * ncot + cotile) * tiles + tile) * ksplit + split
    int tiles;    // pixel tiles of one plane
    int ncot;     // 64-channel output tiles
    int ksplit;   // >= 1
    int cps;      // 32-channel chunks per split
    int nmain;    // Cin / 32
    int n2;       // Gram backward: Cout / 32 chunks of F against the one-tap bank of D, else 0
    int groups;   // workgroups of the launch (a multiple of 8)
    int tiles_y;  // tile rows of one plane
    int d_split, d_tx, d_ty, d_cot, d_n;  // groups / 8 (a workgroup's stride through its XCD's items) as digits of the item index
    int order;    // 0: item = ((n ncot + cotile) tiles + tile) ksplit + split;  1 (round 6): ((n tiles + tile) ncot + cotile) ksplit + split -
                  // the channel tiles of ONE pixel tile side by side on the CUs of an XCD, so that they stage the same patch at the same time
                  // (an XCD's band then holds whole pixel tiles: the input plane leaves memory once instead of once per channel tile)
    float w_inv_scale;
};

#ifdef XP_STAMP
#define XP_MARK(k)                                                                                                       \
    do {                                                                                                                 \
        if (lane == 0 && p.mask && chunk_no < 60) {                                                                      \
            const_cast<float*>(p.mask)[(((int64_t)blockIdx.x * 8 + wave) * 64 + chunk_no) * 8 + (k)] =                   \
                __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());                                       \
        }                                                                                                                \
    } while (0)
#else
#define XP_MARK(k) do {} while (0)
#endif

#define XP_FENCE() __builtin_amdgcn_sched_barrier(0)

// One scheduling region of the K loop = 12 MFMAs + what rides along: per MFMA at most one LDS read, one vector-memory load, one
// vector-memory store, one LDS write and VA vector-ALU instructions, in that order (conv_x3q.hip, XQ_PIPE).
#define XP_PIPE_PLAIN(VA)                                             \
    do {                                                              \
        _Pragma("unroll") for (int m_ = 0; m_ < 12; ++m_) {           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x002, VA, 0);       \
        }                                                             \
    } while (0)
#define XP_PIPE(VA)                                                   \
    do {                                                              \
        _Pragma("unroll") for (int m_ = 0; m_ < 12; ++m_) {           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        \
            __builtin_amdgcn_sched_group_barrier(0x002, VA, 0);       \
        }                                                             \
    } while (0)

// OM: the result is zeroed where p.omask <= 0.  POOL: the epilogue applies ReLU and the 2x2 / 2 max pool behind the layer - `y` is the
// pooled map, p.pool_codes its decision bytes.  UNPOOL: `x` is the POOLED map of a 2x2 / 2 max pool and p.in_codes its decision bytes;
// the input the convolution sees is the pool's backward pass over them, rebuilt while staging.  GRAM: p.dbank / p.dinv / p.omask = the
// one-tap bank of D, its inverse scale and the feature map F: the sums take D . F along (conv_x3w.hip's layout of the bank).
// A launch with p.ksplit > 1 (never POOL or OM) leaves raw partial sums in p.ws for conv_splitk_finish.
template <bool OM, bool POOL, bool UNPOOL, bool GRAM>
__global__ void __launch_bounds__(XP_THREADS, 2) conv_x3p_kernel(ConvArgs p, X3pArgs q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Pl = smem;                    // [part][octet][pos][16 B]
    unsigned char* Wl = smem + XP_PATCH_BYTES;   // [tap][part][octet][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + XP_PATCH_BYTES + XP_W_BYTES);  // per-wave maxima of the chunk being staged
    float* Bl = Ml + 16;                         // bias of every output channel (zeros: none, or split-K)
    unsigned* Tl = reinterpret_cast<unsigned*>(Bl + XP_BIAS_MAX);  // [3][thread]: what a thread's staging items are (set_tile reads them back:
                                                 // registers that live across the K loop for one use per item are registers the loop spills)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, oct = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // (what the K loop needs of the thread's coordinates once per chunk or item is worked out where it is used, from the thread number
    //  laundered through an empty asm: hoisted out of the loop it would be a register the loop spills to scratch - and a scratch reload
    //  is a vector-memory operation that waits behind every store in flight)
    auto my_tid = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    const float w_inv_scale = q.w_inv_scale;
    const int ksplit = q.ksplit;
    const bool slab = ksplit > 1;
    const int in_plane = p.H * p.W;
    const int out_plane = p.OH * p.OW;
    const int st_w = UNPOOL ? p.W >> 1 : p.W;                              // row pitch and plane of the array the patch is staged from
    const int st_plane = UNPOOL ? (p.H >> 1) * (p.W >> 1) : in_plane;
    const unsigned code_mask = (unsigned)p.in_code_mask;

    // This workgroup's items: XCD k (workgroups k, k + 8, ...) owns the k-th contiguous eighth of the list, and its workgroups take those
    // items IN TURN (workgroup j of the XCD: items j, j + groups / 8, ...): at any time the workgroups of an XCD are on horizontally adjacent
    // tiles, so the cache lines a 34-pixel patch row shares with its neighbours (4 bytes of the line to the left, 4 of the line to the right)
    // are asked for by two or three CUs of one L2 at about the same time and leave memory once.  (A contiguous run of tiles per workgroup
    // reads the same patches at half the rate - tools/mfma_probe/stage_bw.hip, 64 channels @ 1024 x 1024: 114 us against 57 us.)
    const int stride = q.groups >> 3;
    const int band_begin = (int)((int64_t)(blockIdx.x & 7) * q.items / 8), band_end = (int)((int64_t)((blockIdx.x & 7) + 1) * q.items / 8);
    const int gq = (int)(blockIdx.x & 7) * stride + (int)(blockIdx.x >> 3);
    const int item_begin = band_begin + (int)(blockIdx.x >> 3), item_end = band_end;
    if (item_begin >= item_end) return;
    if (XP_STAGGER > 0)
        for (int i = 0; i < (gq & 3) * XP_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
    struct Item {
        int n, cot, split, x0, y0, cb, ce;  // image, output-channel tile, K split, pixel origin, chunk range [cb, ce)
    };
    const int tiles_w = p.tiles_x * 32, tiles_h = (q.tiles / p.tiles_x) * XP_ROWS;
    auto chunk_range = [&](Item& it) {
        it.cb = it.split * q.cps;
        it.ce = min(q.nmain, it.cb + q.cps);
    };
    auto decode = [&](int idx) {  // (divisions: once per workgroup)
        Item it;
        int t = idx;
        it.split = t % ksplit;
        t /= ksplit;
        int tile;
        if (q.order) {
            it.cot = t % q.ncot;
            t /= q.ncot;
            tile = t % q.tiles;
            it.n = t / q.tiles;
        } else {
            tile = t % q.tiles;
            t /= q.tiles;
            it.cot = t % q.ncot;
            it.n = t / q.ncot;
        }
        it.x0 = (tile % p.tiles_x) * 32;
        it.y0 = (tile / p.tiles_x) * XP_ROWS;
        it.n = __builtin_amdgcn_readfirstlane(it.n);
        it.cot = __builtin_amdgcn_readfirstlane(it.cot);
        it.split = __builtin_amdgcn_readfirstlane(it.split);
        it.x0 = __builtin_amdgcn_readfirstlane(it.x0);
        it.y0 = __builtin_amdgcn_readfirstlane(it.y0);
        chunk_range(it);
        return it;
    };
    auto advance = [&](Item it) {  // the workgroup's next item: + groups / 8 in the item index - scalar digit additions with carries, no division
        int c;
        it.split += q.d_split;
        c = it.split >= ksplit ? 1 : 0;
        it.split -= c ? ksplit : 0;
        if (q.order) {  // (digits: split, channel tile, tile column, tile row, image)
            it.cot += q.d_cot + c;
            c = it.cot >= q.ncot ? 1 : 0;
            it.cot -= c ? q.ncot : 0;
        }
        it.x0 += (q.d_tx + c) * 32;
        c = it.x0 >= tiles_w ? 1 : 0;
        it.x0 -= c ? tiles_w : 0;
        it.y0 += (q.d_ty + c) * XP_ROWS;
        c = it.y0 >= tiles_h ? 1 : 0;
        it.y0 -= c ? tiles_h : 0;
        if (!q.order) {  // (digits: split, tile column, tile row, channel tile, image)
            it.cot += q.d_cot + c;
            c = it.cot >= q.ncot ? 1 : 0;
            it.cot -= c ? q.ncot : 0;
        }
        it.n += q.d_n + c;
        chunk_range(it);
        return it;
    };

    // Staging items of this thread (plain): the patch position `tid` of each of the chunk's four octets (items 0-3: ONE vector offset,
    // the octet goes into the scalar offset of the load), and - threads 0 ... 399 - position 512 + tid % 100 of octet tid / 100 (item 4);
    // an item's LDS slot is [octet][position][16 B].  voff = byte offset of the position in its plane (item 4: + its octet's planes);
    // out-of-image positions and the missing fifth items get an offset beyond the buffer's range, for which a buffer load returns 0
    // (no selects on the values).  The offsets describe the tile of the chunk BEING STAGED (the next one); what depends on the thread
    // only - patch row and column of its positions - is worked out once (the divisions) and kept packed in two registers.
    //
    // UNPOOL: two items per thread, and they are POOLED elements - (octet, pooled row, pooled column) number tid + 512 k of the 4 x 10 x 18
    // pooled elements whose 2x2 windows cover the 18 x 34 patch (conv_x3q.hip).
    constexpr int NI = UNPOOL ? 2 : XP_NI;
    constexpr int NV = 2;               // vector offsets: plain - [0] items 0-3, [1] item 4; UNPOOL - one per item
    unsigned voff[NV];
    unsigned lds_w[2] = {0u, 0u};       // UNPOOL: the slot of corner 0 of each item
    unsigned vcode[UNPOOL ? 2 : 1], inmask[UNPOOL ? 2 : 1];
    if constexpr (UNPOOL) {
        constexpr int PR = XP_ROWS / 2 + 2, PCW = 18;
#pragma unroll
        for (int k = 0; k < 2; ++k) {  // pooled row | pooled column << 8 | octet << 16 | valid << 24
            const int idx = tid + XP_THREADS * k;
            const int o = idx / (PR * PCW);
            const int rem = idx - o * PR * PCW;
            const int pr = rem / PCW, pc = rem - pr * PCW;
            Tl[k * XP_THREADS + tid] = (unsigned)(pr | pc << 8 | o << 16 | (idx < 4 * PR * PCW ? 1 << 24 : 0));
        }
    } else {  // patch row | column << 8 | octet << 16 | valid << 24 of items 0-3 and of item 4; item 4's LDS slot
        const int ra = tid / XP_PC, ca = tid - ra * XP_PC;
        Tl[tid] = (unsigned)(ra | ca << 8 | 1 << 24);
        const int o4 = tid / 100, pb = 512 + tid - o4 * 100;
        const int rb = pb / XP_PC, cb4 = pb - rb * XP_PC;
        Tl[XP_THREADS + tid] = (unsigned)(rb | cb4 << 8 | o4 << 16 | (tid < 400 ? 1 << 24 : 0));
        vcode[0] = inmask[0] = 0;
    }
    // (a thread reads back its own words only: no barrier needed for them)
    auto set_tile = [&](int x0, int y0) {
        const int tid = my_tid();
        if constexpr (UNPOOL) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const unsigned wh = Tl[k * XP_THREADS + tid];
                const int pr = (int)(wh & 0xffu), pc = (int)((wh >> 8) & 0xffu), o = (int)((wh >> 16) & 0xffu);
                const bool item = (wh >> 24) != 0;
                const int ppy = ((y0 - p.pad) >> 1) + pr, ppx = ((x0 - p.pad) >> 1) + pc;  // (arithmetic shifts: floor for the -1 / -2 of the first tiles)
                const bool ok = item & ((unsigned)ppy < (unsigned)(p.H >> 1)) & ((unsigned)ppx < (unsigned)st_w);
                const int pidx = ppy * st_w + ppx;
                voff[k] = ok ? (unsigned)(o * 8 * st_plane + pidx) * 4u : XP_OOB;
                vcode[k] = ok ? (unsigned)(o * st_plane + pidx) * 8u : XP_OOB;
                const int r0 = 2 * ppy - (y0 - p.pad), c0 = 2 * ppx - (x0 - p.pad);
                inmask[k] = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int r = r0 + (qd >> 1), c = c0 + (qd & 1);
                    inmask[k] |= (item & ((unsigned)r < (unsigned)XP_PR) & ((unsigned)c < (unsigned)XP_PC)) ? 1u << qd : 0u;
                }
                lds_w[k] = (unsigned)(o * XP_PLANE + (r0 * XP_PC + c0) * 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned wh = Tl[k * XP_THREADS + tid];
                const int r = (int)(wh & 0xffu), c = (int)((wh >> 8) & 0xffu), o = (int)((wh >> 16) & 0xffu);
                const int iy = y0 + r - p.pad, ix = x0 + c - p.pad;
                // (unsigned compares: one per extent, no short-circuit branches)
                const bool ok = ((wh >> 24) != 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned v = ok ? (unsigned)(o * 8 * in_plane + iy * p.W + ix) * 4u : XP_OOB;
                if (k == 0) voff[0] = v;
                else Tl[2 * XP_THREADS + tid] = v;  // (item 4's offset waits in LDS: the chunk reads it back for its eight loads)
            }
        }
    };
    const unsigned range = (unsigned)st_plane * 128u;  // 32 planes from the chunk's first one: the range check sees the vector offset only
    float rp[NI][8];       // the next chunk's patch: raw values, then (in place) the packed fp16 halves [0..3] high, [4..7] low
    u32x2 cd[NI];          // UNPOOL: the items' decision bytes
    unsigned sel2[NI][4];  // UNPOOL: the decisions two per register
    struct PatchSrc {
        __amdgpu_buffer_rsrc_t x, codes;
    };
    auto patch_rsrc = [&](int n, int ch) {
        asm volatile("" : "+s"(ch));
        PatchSrc r;
        const int64_t first = ((int64_t)n * p.Cin + (int64_t)ch * 32) * st_plane;
        r.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + first), 0, range, 0x00020000);
        r.codes = r.x;
        if constexpr (UNPOOL) r.codes = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.in_codes + first), 0, range >> 2, 0x00020000);
        return r;
    };
    auto load_patch_part = [&](const PatchSrc& rs, int c_lo, int c_hi) {
        if constexpr (UNPOOL) {
            if (c_lo == 0) {
#pragma unroll
                for (int k = 0; k < NI; ++k) cd[k] = __builtin_amdgcn_raw_buffer_load_b64(rs.codes, vcode[k], 0, 0);
            }
#pragma unroll
            for (int c = c_lo; c < c_hi; ++c)
#pragma unroll
                for (int k = 0; k < NI; ++k)
                    rp[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs.x, voff[k], c * st_plane * 4, 0));
        } else {
#pragma unroll
            for (int c = c_lo; c < c_hi; ++c)
#pragma unroll
                for (int k = 0; k < NI; ++k)  // (items 0-3: the octet's eight planes in the scalar offset)
                    rp[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs.x, k < 4 ? voff[0] : voff[1], ((k < 4 ? 8 * k : 0) + c) * st_plane * 4, 0));
        }
    };
    auto publish_max = [&]() {
        if constexpr (UNPOOL) {
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
        if ((my_tid() & 63) == 0) Ml[wv] = m;
    };
    float sx = 1.f;
    auto chunk_scale = [&]() {
        const float m = fmaxf(fmaxf(fmaxf(Ml[0], Ml[1]), fmaxf(Ml[2], Ml[3])), fmaxf(fmaxf(Ml[4], Ml[5]), fmaxf(Ml[6], Ml[7])));
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;  // floor(log2 m) for normal m
        e = m > 0.f ? max(e, -100) : 11;
        e = __builtin_amdgcn_readfirstlane(e);  // (every lane read the same eight words)
        sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
        return __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23);
    };
    auto split_item = [&](int k) {
        if (k >= NI) return;
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rp[k][c]));
        unsigned hu[4], lu[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float v0 = rp[k][2 * qd], v1 = rp[k][2 * qd + 1];
            const f16x2 h2 = {(_Float16)(v0 * sx), (_Float16)(v1 * sx)};
            const f16x2 l2 = {(_Float16)fmaf(v0, sx, -(float)h2[0]), (_Float16)fmaf(v1, sx, -(float)h2[1])};
            hu[qd] = __builtin_bit_cast(unsigned, h2);
            lu[qd] = __builtin_bit_cast(unsigned, l2);
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            rp[k][qd] = __builtin_bit_cast(float, hu[qd]);
            rp[k][4 + qd] = __builtin_bit_cast(float, lu[qd]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rp[k][c]));
    };
    auto store_patch = [&]() {
        if constexpr (UNPOOL) {
#pragma unroll
            for (int k = 0; k < NI; ++k)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if ((inmask[k] >> qd) & 1u) {
                        const unsigned dst = lds_w[k] + (unsigned)(((qd >> 1) * XP_PC + (qd & 1)) * 16);
                        u32x4 h, l;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned t = sel2[k][i] ^ ((unsigned)qd * 0x00010001u);
                            const unsigned keep = ((t & 0xffffu) ? 0u : 0xffffu) | ((t >> 16) ? 0u : 0xffff0000u);
                            h[i] = __builtin_bit_cast(unsigned, rp[k][i]) & keep;
                            l[i] = __builtin_bit_cast(unsigned, rp[k][4 + i]) & keep;
                        }
                        *reinterpret_cast<u32x4*>(Pl + dst) = h;
                        *reinterpret_cast<u32x4*>(Pl + 4 * XP_PLANE + dst) = l;
                    }
            return;
        }
        const int t_ = my_tid();
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const u32x4 h = {__builtin_bit_cast(unsigned, rp[k][0]), __builtin_bit_cast(unsigned, rp[k][1]), __builtin_bit_cast(unsigned, rp[k][2]),
                             __builtin_bit_cast(unsigned, rp[k][3])};
            const u32x4 l = {__builtin_bit_cast(unsigned, rp[k][4]), __builtin_bit_cast(unsigned, rp[k][5]), __builtin_bit_cast(unsigned, rp[k][6]),
                             __builtin_bit_cast(unsigned, rp[k][7])};
            unsigned dst = (unsigned)t_ * 16u + (unsigned)(XP_PLANE * k);
            if (k == NI - 1) {  // item 4: position 512 + tid % 100 of octet tid / 100; threads without one write zeros into a padding slot
                const unsigned wh = Tl[XP_THREADS + t_];
                dst = (wh >> 24) ? ((wh >> 16) & 0xffu) * XP_PLANE + ((wh & 0xffu) * XP_PC + ((wh >> 8) & 0xffu)) * 16u : (unsigned)(3 * XP_PLANE + (XP_NPOS_PAD - 1) * 16);
            }
            *reinterpret_cast<u32x4*>(Pl + dst) = h;   // (threads without a fifth item write zeros into a padding slot)
            *reinterpret_cast<u32x4*>(Pl + 4 * XP_PLANE + dst) = l;
        }
    };

    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    const unsigned lane16 = lane * 16;
    // Filter slice of a chunk = 72 planes of 1 KiB in LDS order (8 per tap); wave w streams planes first + w, first + w + 8, ...
    auto dma_filters = [&](int ch, int cot, int first, int count, int i0 = 0) {
        const unsigned char* src = bank + ((int64_t)ch * q.ncot + cot) * XP_W_BYTES;
#pragma unroll
        for (int i = i0; i < i0 + count; ++i) {
            const int pl = first + wv + 8 * i;
            const unsigned char* g = src + pl * 1024;
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(Wl + pl * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(g)
                         : "memory");
        }
    };

    int b_base = oct * XP_PLANE + ((2 * wave) * XP_PC + px) * 16;  // (worked out again at the top of every chunk: see my_tid)
    int a_base = oct * 1024 + px * 16;

    f32x4 acc[4][4], master[4][4];  // [16-channel group of the tile][pixel group: row g / 2, column half g % 2]
    f16x8 bf[4][2], af[2][2];       // patch fragments [pixel group][part]; filter fragments [buffer][part]
    auto load_bg = [&](int g, int part, int tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        bf[g][part] = *reinterpret_cast<const f16x8*>(Pl + b_base + part * 4 * XP_PLANE + (((g >> 1) + ky) * XP_PC + (g & 1) * 16 + kx) * 16);
    };
    auto load_ai = [&](int buf, int tap, int i) {
#pragma unroll
        for (int part = 0; part < 2; ++part) af[buf][part] = *reinterpret_cast<const f16x8*>(Wl + a_base + (tap * 2 + part) * 4096 + i * 256);
    };
    auto step_ = [&](auto stores, int tap, int i, int next_tap, auto fresh, float inv, auto va, auto&& extra) {
        constexpr bool FRESH = decltype(fresh)::value;
        constexpr int VA = decltype(va)::value;
        const int cur = i & 1;
        const bool refresh = i == 3 && next_tap >= 0;
        XP_FENCE();
        if (i < 3) load_ai(cur ^ 1, tap, i + 1);
        else if (next_tap >= 0 && tap != 4) load_ai(cur ^ 1, next_tap, 0);  // (tap 5's planes are in LDS behind XM only)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FRESH) {
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(acc[i][g]));
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
        if constexpr (decltype(stores)::value) XP_PIPE(VA);
        else XP_PIPE_PLAIN(VA);
        XP_FENCE();
    };
    auto nothing = []() {};
    constexpr std::true_type FOLD{};
    constexpr std::false_type KEEP{};
    constexpr std::integral_constant<int, 0> V0{};
    constexpr std::integral_constant<int, 2> V2{};
    constexpr std::integral_constant<int, 4> V4{};

    Item cur = decode(item_begin);
    // ---- the finished item's epilogue, in sixteen pieces (plain / masked / slab: piece k = group k / 4, pixel group k % 4) or eight
    //      (POOL: group k / 2, column half k % 2); `fin` = that item, set when its last chunk has run
    Item fin = cur;
    bool fin_valid = false;                  // (the first item's first chunk carries the epilogue of nothing: every offset out of range)
    __amdgpu_buffer_rsrc_t rs_out, rs_aux;   // the output array (image `fin.n`) and the mask (OM) / the decision bytes (POOL)
    unsigned so_base = 0;                    // scalar byte offset of the tile's first element of the tile's first channel
    unsigned vo[2];                          // the lane's byte offset for column half h (rows and channels go into the scalar offset)
    bool row_ok[4] = {true, true, true, true};
    const int PW = p.OW >> 1;
    const int pplane = (p.OH >> 1) * PW;
    auto epi_begin = [&]() {
        const int t_ = my_tid();
        const int lane = t_ & 63, px = t_ & 15, oct = (t_ >> 4) & 3, wave = wv;
        if constexpr (POOL) {
            const int64_t img = (int64_t)fin.n * p.Cout * pplane;
            rs_out = __builtin_amdgcn_make_buffer_rsrc(p.y + img, 0, (unsigned)p.Cout * (unsigned)pplane * 4u, 0x00020000);
            rs_aux = __builtin_amdgcn_make_buffer_rsrc(p.pool_codes + img, 0, (unsigned)p.Cout * (unsigned)pplane, 0x00020000);
            so_base = (unsigned)((fin.y0 >> 1) * PW + (fin.x0 >> 1));  // in pooled pixels
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int oy = fin.y0 + 2 * wave, oxx = fin.x0 + 16 * h + px;
                const bool store = fin_valid && (lane & 1) == 0 && oy + 1 < p.OH && oxx + 1 < p.OW;
                vo[h] = store ? (unsigned)(wave * PW + ((16 * h + px) >> 1)) : XP_OOB;  // pooled pixel inside the tile (x 4 / x 8 when used)
            }
        } else {
            const int64_t img = (int64_t)fin.n * p.Cout * out_plane;
            float* base = slab ? p.ws + ((int64_t)fin.n * ksplit + fin.split) * p.Cout * out_plane : p.y + img;
            rs_out = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)p.Cout * (unsigned)out_plane * 4u, 0x00020000);
            rs_aux = rs_out;
            if constexpr (OM) rs_aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.omask + img), 0, (unsigned)p.Cout * (unsigned)out_plane * 4u, 0x00020000);
            so_base = (unsigned)((fin.cot * XP_COT) * out_plane + fin.y0 * p.OW + fin.x0) * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                vo[h] = fin_valid && fin.x0 + 16 * h + px < p.OW ? (unsigned)((4 * oct) * out_plane + (2 * wave) * p.OW + 16 * h + px) * 4u : XP_OOB;
#pragma unroll
            for (int g = 0; g < 4; ++g) row_ok[g] = fin.y0 + 2 * wv + (g >> 1) < p.OH;
        }
    };
    float mk[2][4];  // OM: the mask values of the pieces in flight, [piece % 2][r]
    auto epi_load = [&](int k) {  // masks of piece k
        if constexpr (OM) {
            const int i = k >> 2, g = k & 3;
            const unsigned v = row_ok[g] ? vo[g & 1] : XP_OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mk[k % 2][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs_aux, v, so_base + (unsigned)((16 * i + r) * out_plane + (g >> 1) * p.OW) * 4u, 0));
        }
    };
    // the new item's starting values of group i: its bias (LDS table; zeros where there is none)
    auto reset_group = [&](int i, int cot) {
        const int oct = (my_tid() >> 4) & 3;
        const f32x4 b = *reinterpret_cast<const f32x4*>(Bl + cot * XP_COT + 16 * i + 4 * oct);
#pragma unroll
        for (int g = 0; g < 4; ++g) master[i][g] = b;
    };
    auto epi_piece = [&](int k, int next_cot) {
        const int oct = (my_tid() >> 4) & 3;
        if constexpr (POOL) {
            if (k >= 8) return;
            const int i = k >> 1, h = k & 1;
            unsigned pk = 0;
            const unsigned so = (unsigned)((fin.cot * XP_COT + 16 * i) * pplane) + so_base;  // pooled pixels from channel 0's first
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
                // channel 16 i + 4 oct + r: the lane's share of the offset is (4 oct) planes + its pooled pixel
                const unsigned v = vo[h] == XP_OOB ? XP_OOB : ((unsigned)(4 * oct * pplane) + vo[h]) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rs_out, v, (so + (unsigned)(r * pplane)) * 4u, 0);
            }
            // decision bytes: [octet of channels][pooled pixel][8 channels]; the lane's four channels are one dword
            const unsigned vc = vo[h] == XP_OOB ? XP_OOB : ((unsigned)((oct >> 1) * pplane) + vo[h]) * 8u + 4u * (oct & 1);
            __builtin_amdgcn_raw_buffer_store_b32(pk, rs_aux, vc, ((unsigned)(((fin.cot * XP_COT + 16 * i) >> 3) * pplane) + so_base) * 8u, 0);
            if (h == 1) reset_group(i, next_cot);
        } else {
            const int i = k >> 2, g = k & 3;
            const unsigned v = row_ok[g] ? vo[g & 1] : XP_OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float val = master[i][g][r];
                if (p.relu && !slab) val = val > 0.f ? val : 0.f;
                if constexpr (OM) val = mk[k % 2][r] > 0.f ? val : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_out, v,
                                                      so_base + (unsigned)((16 * i + r) * out_plane + (g >> 1) * p.OW) * 4u, 0);
            }
            if (g == 3) reset_group(i, next_cot);
        }
    };

    // ---- Gram backward of the finished item's tile (GRAM): the 32-channel chunks of F against the matching columns of D, one tap,
    //      48 MFMAs per chunk and wave, nothing through LDS: a lane's operands are what it can load itself - its four pixel groups for
    //      its octet's 8 channels (B), its output channel's 8 values of D per part and channel group (A, from conv_x3w.hip's bank:
    //      [chunk16][cotile][part][octet 2][co 64][8 ch] = 4 KiB per (chunk, tile)) - and the power-of-two scale is per WAVE.
    auto gram_phase = [&](const Item& it, float inv_pending) {
        // the sums of the last chunk's taps 5-8 first: the accumulators start again from zero
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    master[i][g][r] = fmaf(acc[i][g][r], inv_pending, master[i][g][r]);
                    acc[i][g][r] = 0.f;
                }
        const float d_inv_scale = p.dinv[it.n];
        const unsigned char* dbank = reinterpret_cast<const unsigned char*>(p.dbank);
        unsigned vf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int oy = it.y0 + 2 * wave + (g >> 1), oxx = it.x0 + (g & 1) * 16 + px;
            vf[g] = (oy < p.OH && oxx < p.OW) ? (unsigned)((oct * 8) * out_plane + oy * p.OW + oxx) * 4u : XP_OOB;
        }
        const unsigned frange = (unsigned)out_plane * 128u;
        float f[4][8];
        auto request = [&](int c2) {
            asm volatile("" : "+s"(c2));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.omask + ((int64_t)it.n * p.Cout + (int64_t)c2 * 32) * out_plane), 0, frange, 0x00020000);
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) f[g][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vf[g], c * out_plane * 4, 0));
        };
        request(0);
        for (int c2 = 0; c2 < q.n2; ++c2) {
            float m = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(f[g][c]));
            m = wave_max_nonneg(m);
            int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
            e = m > 0.f ? max(e, -100) : 11;
            const float s = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
            const float inv = __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23) * d_inv_scale;
            f16x8 b[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 Hh, Ll;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float v0 = f[g][2 * qd], v1 = f[g][2 * qd + 1];
                    const f16x2 h2 = {(_Float16)(v0 * s), (_Float16)(v1 * s)};
                    const f16x2 l2 = {(_Float16)fmaf(v0, s, -(float)h2[0]), (_Float16)fmaf(v1, s, -(float)h2[1])};
                    Hh[qd] = __builtin_bit_cast(unsigned, h2);
                    Ll[qd] = __builtin_bit_cast(unsigned, l2);
                }
                b[g][0] = __builtin_bit_cast(f16x8, Hh);
                b[g][1] = __builtin_bit_cast(f16x8, Ll);
            }
            // D: chunk16 = 2 c2 + (oct >> 1), octet = oct & 1, co = 16 i + px
            const unsigned char* gsrc = dbank + ((((int64_t)it.n * (2 * q.n2) + 2 * c2 + (oct >> 1)) * q.ncot + it.cot) * 4096) + (oct & 1) * 1024 + px * 16;
            u32x4 a[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int part = 0; part < 2; ++part) a[i][part] = *reinterpret_cast<const u32x4*>(gsrc + part * 2048 + i * 256);
            if (c2 + 1 < q.n2) request(c2 + 1);  // (the raw values are split: their registers take the next chunk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 ah = __builtin_bit_cast(f16x8, a[i][0]), al = __builtin_bit_cast(f16x8, a[i][1]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b[g][0], acc[i][g], 0, 0, 0);  // smallest terms first
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b[g][1], acc[i][g], 0, 0, 0);
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b[g][0], acc[i][g], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        master[i][g][r] = fmaf(acc[i][g][r], inv, master[i][g][r]);
                        acc[i][g][r] = 0.f;
                    }
        }
    };

    // ---- start: bias table, the first item's first chunk (conv_x3q.hip's prologue)
    for (int c = tid; c < p.Cout; c += XP_THREADS) Bl[c] = (p.bias != nullptr && !slab) ? p.bias[c] : 0.f;
    set_tile(cur.x0, cur.y0);
    int ch = cur.cb;
    dma_filters(ch, cur.cot, 0, 9);
    if constexpr (!UNPOOL) voff[1] = Tl[2 * XP_THREADS + tid];
    load_patch_part(patch_rsrc(cur.n, ch), 0, 8);
    publish_max();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float inv_prev = 0.f, inv_cur = 0.f, inv_next = 0.f;  // un-scaling factors: previous chunk (its taps 5-8 wait in acc), this one, the next
    inv_cur = chunk_scale() * w_inv_scale;
    inv_next = inv_cur;
#pragma unroll
    for (int k = 0; k < NI; ++k) split_item(k);
    store_patch();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        reset_group(i, cur.cot);  // (the table is complete behind the barrier above)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#ifdef XP_STAMP
    int chunk_no = 0;
#endif
    // One chunk (conv_x3q.hip's schedule; step s = 4 tap + group, 36 per chunk):
    //   tap 0 [fold of the previous chunk's sums]  taps 0-3 [patch(next) requested; the planes of taps 5-8 of THIS chunk]
    //   tap 4 [maximum of patch(next) -> LDS]   | XM |   scale(next),  tap 5 [fold of taps 0-4]  taps 5-8 [planes of taps 0-4 (next);
    //   patch(next) split between the MFMAs]   | X1 |   patch(next) -> LDS   | X2 |
    // EPI: the first chunk of an item, with the previous item's epilogue: its masters are final behind tap 0's folds and not needed
    // before the NEXT chunk's tap 0 - this chunk keeps all nine taps in the accumulators (no fold at tap 5) - so the pieces spread over
    // the whole chunk, behind everything the chunk has to wait for: loads and LDS-DMA go out in steps 0-7 and 20-24, stores in steps
    // 8-16 and 25-31, and the waits count the stores issued since (memory operations return in issue order: `vmcnt(n)` = all but the n
    // youngest are done), so no wait of the chunk's own pipeline waits for a store.  (Masked form, OM: the pieces' mask loads interleave with
    // the stores, so `vmcnt(ST1 / ST2)` does wait for the stores issued before the youngest mask loads - conservative, still correct.)
    auto chunk_body = [&](auto epi_tag, const Item& it, int chn, bool later, bool more, const Item& nx, int nch, const PatchSrc& rs) {
        constexpr bool EPI = decltype(epi_tag)::value;
        auto step = [&](int tap, int i, int next_tap, auto fresh, float inv, auto va, auto&& extra) {
            step_(std::integral_constant<bool, EPI>{}, tap, i, next_tap, fresh, inv, va, extra);
        };
        {
            const int t_ = my_tid();
            if constexpr (!UNPOOL) voff[1] = Tl[2 * XP_THREADS + t_];
            b_base = ((t_ >> 4) & 3) * XP_PLANE + ((2 * wv) * XP_PC + (t_ & 15)) * 16;
            a_base = ((t_ >> 4) & 3) * 1024 + (t_ & 15) * 16;
        }
        const int ncot_ = it.cot;  // (the running item's tile: what the masters restart from)
        if constexpr (EPI) epi_begin();
        // pieces: plain / masked 16, pooling 8; the first XP_W1_PIECES (pooling: half as many) end with step 16, the others start with step 20
        // (pooling: every other step); a masked piece's mask values are requested one piece ahead (behind the previous piece's use of the other slot)
        constexpr int NP = POOL ? 8 : 16;
        constexpr int P1 = xp_first_pieces(POOL);
        auto step_of = [](int k) { return xp_step_of(POOL, k); };
        auto E = [&](int s) {
            if constexpr (EPI) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if constexpr (OM) {
                        if (k < 1 ? s == step_of(0) - 1 : s == step_of(k - 1)) epi_load(k);
                    }
                }
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (s == step_of(k) && fin_valid) epi_piece(k, ncot_);
            }
        };
        // stores issued behind the chunk's last load / LDS-DMA of each half (what the waits may leave in flight)
        constexpr int SPP = POOL ? 5 : 4;                                // stores per piece
        constexpr int ST1 = !EPI ? 0 : SPP * P1;
        constexpr int late2 = POOL ? (NP - P1 > 3 ? NP - P1 - 3 : 0) : (NP - P1 > 5 ? NP - P1 - 5 : 0);  // pieces behind step 24's LDS-DMA
        constexpr int ST2 = !EPI ? 0 : SPP * late2;
        static_assert(ST1 <= 60 && ST2 <= 60 && xp_step_of(POOL, NP - 1) <= 31 && xp_step_of(POOL, 0) >= 5, "conv_x3p: epilogue schedule");
        constexpr std::integral_constant<int, EPI ? 4 : 2> VF{};
        constexpr std::integral_constant<int, EPI ? 2 : 0> VK{};
        XP_MARK(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            load_bg(g, 0, 0);
            load_bg(g, 1, 0);
        }
        load_ai(0, 0, 0);
        if constexpr (EPI || XP_EARLY_LOADS) {
            step(0, 0, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 0, 1); });
            step(0, 1, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 1, 2); if (later) dma_filters(chn, it.cot, 40, 1, 0); });
            step(0, 2, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 2, 3); });
            step(0, 3, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 3, 4); if (later) dma_filters(chn, it.cot, 40, 1, 1); });
            step(1, 0, 2, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 4, 5); E(4); });
            step(1, 1, 2, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 5, 6); if (later) dma_filters(chn, it.cot, 40, 1, 2); E(5); });
            step(1, 2, 2, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 6, 7); E(6); });
            step(1, 3, 2, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 7, 8); if (later) dma_filters(chn, it.cot, 40, 1, 3); E(7); });
#pragma unroll
            for (int tap = 2; tap < 4; ++tap)
#pragma unroll
                for (int i = 0; i < 4; ++i) step(tap, i, tap + 1, KEEP, 0.f, VK, [&]() { E(4 * tap + i); });
        } else {
            step(0, 0, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 0, 1); });
            step(0, 1, 1, FOLD, inv_prev, VF, nothing);
            step(0, 2, 1, FOLD, inv_prev, VF, [&]() { load_patch_part(rs, 1, 2); });
            step(0, 3, 1, FOLD, inv_prev, VF, [&]() { if (later) dma_filters(chn, it.cot, 40, 1, 0); });
#pragma unroll
            for (int tap = 1; tap < 4; ++tap) {
                step(tap, 0, tap + 1, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 2 * tap, 2 * tap + 1); });
                step(tap, 1, tap + 1, KEEP, 0.f, VK, nothing);
                step(tap, 2, tap + 1, KEEP, 0.f, VK, [&]() { load_patch_part(rs, 2 * tap + 1, 2 * tap + 2); });
                step(tap, 3, tap + 1, KEEP, 0.f, VK, [&]() { if (later) dma_filters(chn, it.cot, 40, 1, tap); });
            }
        }
        XP_MARK(1);
        step(4, 0, 5, KEEP, 0.f, VK, [&]() { E(16); });
        // the patch is in registers (and this wave's planes of taps 5-8 in LDS): everything but the stores issued since - when there were any
        // (the workgroup's first item carries no epilogue: nothing was stored, so the count would leave loads and LDS-DMA in flight)
        if (EPI && fin_valid) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step(4, 1, 5, KEEP, 0.f, V4, [&]() { publish_max(); });
        step(4, 2, 5, KEEP, 0.f, V4, nothing);
        step(4, 3, 5, KEEP, 0.f, V0, nothing);
        XP_MARK(2);
        if (EPI && fin_valid) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // XM: the filter planes of taps 0-4 are free, those of taps 5-8 and the maxima of the next chunk visible
        XP_MARK(3);
        load_ai(0, 5, 0);
        inv_next = chunk_scale() * w_inv_scale;
        if constexpr (EPI) {
            step(5, 0, 6, KEEP, 0.f, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 0); E(20); });
            step(5, 1, 6, KEEP, 0.f, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 1); E(21); });
            step(5, 2, 6, KEEP, 0.f, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 2); E(22); });
            step(5, 3, 6, KEEP, 0.f, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 3); E(23); });
            step(6, 0, 7, KEEP, 0.f, V4, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 4); split_item(0); E(24); });
            step(6, 1, 7, KEEP, 0.f, V4, [&]() { split_item(1); E(25); });
            step(6, 2, 7, KEEP, 0.f, V4, [&]() { split_item(2); E(26); });
            step(6, 3, 7, KEEP, 0.f, V4, [&]() { split_item(3); E(27); });
            step(7, 0, 8, KEEP, 0.f, V4, [&]() { split_item(4); E(28); });
            step(7, 1, 8, KEEP, 0.f, V2, [&]() { E(29); });
            step(7, 2, 8, KEEP, 0.f, V2, [&]() { E(30); });
            step(7, 3, 8, KEEP, 0.f, V2, [&]() { E(31); });
            step(8, 0, -1, KEEP, 0.f, V0, nothing);
            step(8, 1, -1, KEEP, 0.f, V0, nothing);
            step(8, 2, -1, KEEP, 0.f, V0, nothing);
            step(8, 3, -1, KEEP, 0.f, V0, nothing);
        } else {
            constexpr std::integral_constant<bool, XP_MIDFOLD != 0> MID{};
            step(5, 0, 6, MID, inv_cur, V2, nothing);
            step(5, 1, 6, MID, inv_cur, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 0); });
            step(5, 2, 6, MID, inv_cur, V2, nothing);
            step(5, 3, 6, MID, inv_cur, V2, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 1); });
            step(6, 0, 7, KEEP, 0.f, V4, [&]() { split_item(0); });
            step(6, 1, 7, KEEP, 0.f, V4, [&]() { split_item(1); });
            step(6, 2, 7, KEEP, 0.f, V4, [&]() { split_item(2); });
            step(6, 3, 7, KEEP, 0.f, V0, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 2); });
            step(7, 0, 8, KEEP, 0.f, V4, [&]() { split_item(3); });
            step(7, 1, 8, KEEP, 0.f, V4, [&]() { split_item(4); });
            step(7, 2, 8, KEEP, 0.f, V0, nothing);
            step(7, 3, 8, KEEP, 0.f, V0, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 3); });
            step(8, 0, -1, KEEP, 0.f, V0, nothing);
            step(8, 1, -1, KEEP, 0.f, V0, [&]() { if (more) dma_filters(nch, nx.cot, 0, 1, 4); });
            step(8, 2, -1, KEEP, 0.f, V0, nothing);
            step(8, 3, -1, KEEP, 0.f, V0, nothing);
        }
        XP_MARK(4);
        inv_prev = inv_cur;
        inv_cur = inv_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X1: every wave is done reading the patch and the remaining filter planes
        XP_MARK(5);
        store_patch();
        if (EPI && fin_valid) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X2: patch and the filters of taps 0-4 of the next chunk are in LDS
        XP_MARK(6);
#ifdef XP_STAMP
        XP_MARK(7);
        ++chunk_no;
#endif
    };

    // Items in turn: the first chunk of an item carries the previous item's epilogue (the first item's: of nothing - every offset out of
    // range, the masters reset to the values they have), the others are plain; one static instance of each form, no branch between them.
    int item = item_begin;
    bool later = false;
    Item nx = cur;
    int nch = 0;
    bool more = true;
    PatchSrc rs;
    auto plan_next = [&](int chn) {
        const bool last_of_item = chn + 1 >= cur.ce;
        more = !(last_of_item && item + stride >= item_end);
        nx = cur;
        nch = chn + 1;
        if (last_of_item && more) {  // the next chunk opens another item: staging moves to its tile
            nx = advance(cur);
            nch = nx.cb;
            set_tile(nx.x0, nx.y0);
        }
        // (no branch around anything that defines registers: the last chunk of all stages itself once more; only the filter DMA is skipped)
        rs = patch_rsrc(more ? nx.n : cur.n, more ? nch : chn);
    };
    while (true) {
        // (the first chunk of EVERY item runs the form without the fold at tap 5 - the workgroup's first item with no epilogue to carry
        //  skips the pieces - so that a tile's arithmetic does not depend on where in a workgroup's list it sits)
        plan_next(cur.cb);
        chunk_body(std::true_type{}, cur, cur.cb, later, more, nx, nch, rs);
        later = true;
        for (int chn = cur.cb + 1; chn < cur.ce; ++chn) {
            plan_next(chn);
            chunk_body(std::false_type{}, cur, chn, later, more, nx, nch, rs);
        }
        if constexpr (GRAM) {
            if (cur.split == ksplit - 1) {
                gram_phase(cur, inv_prev);
                inv_prev = 0.f;  // (the pending sums are in the masters)
            }
        }
        fin = cur;
        fin_valid = true;
        item += stride;
        if (!more) break;
        cur = nx;
    }
    // the last item's epilogue, on its own
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) master[i][g][r] = fmaf(acc[i][g][r], inv_prev, master[i][g][r]);
    epi_begin();
    if constexpr (POOL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) epi_piece(k, fin.cot);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                epi_load(4 * i + g);
                epi_piece(4 * i + g, fin.cot);
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------

static int x3p_choose_split(const ConvArgs& a, int n) {
    // items = tiles x output-channel tiles x frames; every workgroup walks its share, so a split only pays where the list is shorter than
    // the chip (conv5_1 at 1024 x 1024: 8 x 8 items of 16 chunks)
    const int64_t items = (int64_t)((a.OW + 31) / 32) * ((a.OH + XP_ROWS - 1) / XP_ROWS) * ((a.Cout + XP_COT - 1) / XP_COT) * split_batch_hint();
    (void)n;
    const int nchunks = a.Cin / 32;
    const int forced = (int)tuning("x3p_ks", 0);
    // (every split gets chunks: ks is brought down to the number of non-empty ranges of ceil(nchunks / ks) chunks)
    auto whole = [&](int ks) { const int cps = (nchunks + ks - 1) / ks; return (nchunks + cps - 1) / cps; };
    if (forced > 0) return whole(forced <= nchunks / 2 ? forced : (nchunks >= 4 ? nchunks / 2 : 1));
    if (items >= 512 || nchunks < 4) return 1;
    const double out_mb = (double)split_batch_hint() * a.Cout * a.OH * a.OW * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 2; ++ks) {
        const int64_t its = items * ks;
        const double per_wg = (double)((its + 255) / 256);  // items of the busiest workgroup
        double cost = (per_wg * (double)((nchunks + ks - 1) / ks) + 0.7) * 9.0;  // ~9 us per 32-channel chunk of a full CU
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;
        if (cost < best_cost * 0.97) {
            best_cost = cost;
            best = ks;
        }
    }
    return whole(best);
}

bool conv_x3p_supports(const ConvArgs& a) {
    // 32-bit byte offsets inside one image's output (and the out-of-range marker above them), the bias table in LDS, whole 64-channel tiles
    return a.Cin % 32 == 0 && a.Cout % XP_COT == 0 && a.Cout <= XP_BIAS_MAX && (int64_t)a.H * a.W <= (1ll << 24) && a.pad >= 0 && a.pad <= 2 &&
           (int64_t)a.Cout * a.OH * a.OW * 4 < (1ll << 31) && (int64_t)a.OH * a.OW <= (1ll << 24);
}

// workgroups of a launch: one per CU (tuning constant x3p_groups / maua_conv_x3p_set_max_groups: fewer, so that small test shapes walk many items
// per workgroup; a multiple of 8)
static int g_xp_max_groups = 0;  // (0: the tuning constant x3p_groups, 256 by default)
static int x3p_max_groups() {
    const int v = g_xp_max_groups > 0 ? g_xp_max_groups : (int)tuning("x3p_groups", 256);
    return v >= 8 ? v / 8 * 8 : 256;
}

#ifdef XP_STAMP
static float* g_xp_stamp = nullptr;
extern "C" void maua_xp_set_stamp_buffer(float* buf) { g_xp_stamp = buf; }
#endif

template <bool OM, bool POOL, bool UNPOOL, bool GRAM>
static int xp_launch_one(const ConvArgs& p, const X3pArgs& q, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&conv_x3p_kernel<OM, POOL, UNPOOL, GRAM>);
    static unsigned long long served = 0;  // (once per instantiation and device)
    const hipError_t rc = opt_in_dynamic_lds(fn, XP_LDS_BYTES, &served);
    if (rc != hipSuccess) {
        set_error("conv_x3p: hipFuncSetAttribute: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    hipLaunchKernelGGL((conv_x3p_kernel<OM, POOL, UNPOOL, GRAM>), dim3((unsigned)q.groups), dim3(XP_THREADS), XP_LDS_BYTES, stream, p, q);
    return check_launch("conv_x3p_kernel");
}

int conv_x3p_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
#ifdef XP_STAMP
    p.mask = g_xp_stamp;
#endif
    p.tiles_x = (a.OW + 31) / 32;
    X3pArgs q{};
    q.tiles = p.tiles_x * ((a.OH + XP_ROWS - 1) / XP_ROWS);
    q.ncot = a.Cout / XP_COT;
    const int ks = a.ws ? x3p_choose_split(a, n) : 1;
    q.ksplit = ks;
    p.ksplit = ks;
    q.nmain = a.Cin / 32;
    q.cps = (q.nmain + ks - 1) / ks;
    q.n2 = a.dbank ? a.Cout / 32 : 0;
    const int64_t items = (int64_t)n * q.ncot * q.tiles * ks;
    if (items > (1ll << 30)) {
        set_error("conv_x3p: too many work items");
        return MAUA_E_UNSUPPORTED;
    }
    q.items = (int)items;
    const int max_groups = x3p_max_groups();
    q.groups = (int)(items >= max_groups ? max_groups : (items + 7) / 8 * 8);
    q.tiles_y = q.tiles / p.tiles_x;
    q.order = tuning("x3p_order", 1) != 0 ? 1 : 0;
    {
        int st = q.groups / 8;
        q.d_split = st % ks;
        st /= ks;
        if (q.order) {
            q.d_cot = st % q.ncot;
            st /= q.ncot;
        }
        q.d_tx = st % p.tiles_x;
        st /= p.tiles_x;
        q.d_ty = st % q.tiles_y;
        st /= q.tiles_y;
        if (q.order) {
            q.d_n = st;
        } else {
            q.d_cot = st % q.ncot;
            q.d_n = st / q.ncot;
        }
    }
    q.w_inv_scale = 1.f / w_scale;
    const bool om = ks == 1 && a.omask != nullptr, gram = a.dbank != nullptr, unpool = a.in_codes != nullptr;
    int rc;
    if (a.pool_codes && ks == 1) rc = xp_launch_one<false, true, false, false>(p, q, stream);
    else if (unpool && gram && om) rc = xp_launch_one<true, false, true, true>(p, q, stream);
    else if (unpool && gram) rc = xp_launch_one<false, false, true, true>(p, q, stream);
    else if (unpool && om) rc = xp_launch_one<true, false, true, false>(p, q, stream);
    else if (unpool) rc = xp_launch_one<false, false, true, false>(p, q, stream);
    else if (gram && om) rc = xp_launch_one<true, false, false, true>(p, q, stream);
    else if (gram) rc = xp_launch_one<false, false, false, true>(p, q, stream);
    else if (om) rc = xp_launch_one<true, false, false, false>(p, q, stream);
    else rc = xp_launch_one<false, false, false, false>(p, q, stream);
    if (rc || ks == 1) return rc;
    return a.pool_codes ? conv_splitk_finish_pool(a, n, ks, stream) : conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_conv_x3p_supported(int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(1, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.H = h;
    a.W = w;
    a.Cout = cout;
    a.pad = pad;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    return a.OH > 0 && a.OW > 0 && conv_x3p_supports(a) ? 1 : 0;
}

int maua_conv_x3p_split(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    return x3p_choose_split(a, n);
}

size_t maua_conv_x3p_workspace_bytes(int n, int cin, int h, int w, int cout, int pad) {
    const int ks = maua_conv_x3p_split(n, cin, h, w, cout, pad);
    return ks > 1 ? (size_t)n * ks * cout * (h + 2 * pad - 2) * (w + 2 * pad - 2) * sizeof(float) : 0;
}

int maua_conv_x3p_set_max_groups(int groups) {
    const int before = g_xp_max_groups;  // the raw override (0 = none), so that save / restore is exact
    g_xp_max_groups = groups >= 8 ? groups / 8 * 8 : 0;
    return before;
}

int maua_conv_x3p_preferred(int n, int cin, int h, int w, int cout, int pad) {
    // The host side's routing rule (measured in the network, profiles/probes_r05.md): the persistent kernel pays where a workgroup walks
    // two items and more (a launch of one item per workgroup has nothing to overlap: conv_x3q.hip is the same loop with less bookkeeping),
    // and where the list deals out evenly: what a launch loses is the difference between the busiest workgroup's share and the mean, plus
    // the pixels its 16 x 32 tiles cover beyond the plane.  Below the bounds conv_x3q / conv_x3w run.
    if (!maua_conv_x3p_supported(cin, h, w, cout, pad) || !conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    const int oh = h + 2 * pad - 2, ow = w + 2 * pad - 2;
    const double min_fill = tuning("x3p_min_fill", 0.8);
    const int min_items = (int)tuning("x3p_min_items", 512);
    const int ks = maua_conv_x3p_split(n, cin, h, w, cout, pad);
    const int64_t items = (int64_t)((ow + 31) / 32) * ((oh + XP_ROWS - 1) / XP_ROWS) * (cout / XP_COT) * split_batch_hint() * ks;
    const double fill = (double)items / (double)(((items + 255) / 256) * 256);
    const double cover = (double)oh * ow / ((double)((oh + XP_ROWS - 1) / XP_ROWS * XP_ROWS) * ((ow + 31) / 32 * 32));
    // (a list that is only long enough with the channel loop split is the short-grid case conv_x3q's own split-K form serves as well)
    // ... and a workgroup's share should be eight chunks and more (conv1_2 of a 512 x 512 image - two items of two chunks - loses)
    return ks == 1 && items >= min_items && items * (cin / 32) >= 4 * (int64_t)min_items && fill * cover >= min_fill ? 1 : 0;
}

int maua_conv3x3_x3p(const float* x, const unsigned char* in_codes, int honour_relu_bit, const void* bank, float w_scale, const float* bias,
                     const float* out_relu_mask, const void* dmat_bank, const float* dmat_inv_scale, float* y, unsigned char* pool_codes,
                     int n, int cin, int h, int w, int cout, int pad, int relu, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y && w_scale > 0.f, MAUA_E_INVAL, "conv3x3_x3p: bad args");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x3p: bad dims");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_x3p: input smaller than the filter");
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
    MAUA_REQUIRE(conv_x3p_supports(a), MAUA_E_UNSUPPORTED,
                 "conv3x3_x3p: needs cin %% 32 == 0, cout %% 64 == 0, cout <= 512, planes of at most 2^24 pixels and an output image below 2 GiB");
    if (pool_codes) {
        MAUA_REQUIRE(a.OH >= 2 && a.OW >= 2 && !out_relu_mask && !dmat_bank && !in_codes && relu, MAUA_E_UNSUPPORTED,
                     "conv3x3_x3p: the pooling form is conv + bias + ReLU + pool of an output plane of 2 x 2 and more");
        a.pool_codes = pool_codes;
    }
    if (dmat_bank) {
        MAUA_REQUIRE(dmat_inv_scale && out_relu_mask, MAUA_E_INVAL, "conv3x3_x3p: the Gram term needs the feature map and the bank's inverse scale");
        MAUA_REQUIRE(a.OH == h && a.OW == w, MAUA_E_UNSUPPORTED, "conv3x3_x3p: the Gram term needs an output plane of the input's size");
        a.dbank = dmat_bank;
        a.dinv = dmat_inv_scale;
    }
    if (in_codes) {
        MAUA_REQUIRE(h >= 2 && w >= 2, MAUA_E_UNSUPPORTED, "conv3x3_x3p: the unpooling form needs an input plane of 2 x 2 and more");
        a.in_codes = in_codes;
        a.in_code_mask = honour_relu_bit ? 7 : 3;
    }
    a.ws = (workspace && workspace_bytes >= maua_conv_x3p_workspace_bytes(n, cin, h, w, cout, pad) && maua_conv_x3p_split(n, cin, h, w, cout, pad) > 1)
               ? (float*)workspace : nullptr;
    return conv_x3p_launch(a, n, w_scale, (hipStream_t)stream);
}

}  // extern "C"
