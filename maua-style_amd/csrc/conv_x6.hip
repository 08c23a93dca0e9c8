// 3x3 stride-1 convolution with fp32-level accuracy on the bf16 matrix cores ("bf16x6").
//
// An fp32 value splits exactly into three bf16 parts x = xh + xm + xl (8 significant bits each).  A product of two
// fp32 numbers is then the sum of nine bf16 x bf16 products, each exact in fp32; the six largest
//        xh*wh + (xh*wm + xm*wh) + (xh*wl + xm*wm + xl*wh)
// carry everything down to 2^-24 relative, i.e. fp32 rounding level.  Six v_mfma_f32_32x32x16_bf16 (1024 FLOP/clk/SIMD)
// replace sixteen v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD) worth of work: 2.67x the fp32 matrix rate at the same
// accuracy, with fp32 accumulation inside the MFMA.
//
// Geometry: workgroup = 4 waves = 64 output channels x (4 rows x 32 px); wave = 64 co x one 32-px row (two 32x32
// accumulators).  K is consumed in chunks of 8 input channels; one MFMA k-step (K = 16) = two taps x 8 channels (lane
// half h takes tap 2s+h; the ninth tap runs alone at K = 8, v_mfma_f32_32x32x8_bf16_1k).  LDS holds ONE chunk: three bf16 images of the
// patch [part][pos][8ch] (16 B per position: conflict-free b128 reads and writes) and the filter slice
// [tap][co][part][8ch] (48-byte lane stride: conflict-free) = 37.6 KB, so four workgroups share a CU (4 waves/SIMD)
// and hide each other's barriers.  The filter bank is pre-split on the device once (maua_conv_pack_filters_x6) in
// exactly the LDS order, so a chunk's slice is one contiguous 27 KB block copied by 27 global_load_lds_dwordx4
// (no VGPRs); activations are split while they are staged (v_cvt_pk_bf16_f32 + subtract).
// The ReLU mask of a backward-data pass is NOT applied here while staging; it is applied by the producer of the
// gradient (`omask` in this kernel's epilogue, the pooling backward, the loss kernels).
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>

#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int X6_COT = 64;                    // output channels per workgroup
constexpr int X6_PH = 4, X6_PR = 6, X6_PC = 34;
constexpr int X6_NPOS = X6_PR * X6_PC;        // 204
constexpr int X6_NPOS_PAD = 208;
constexpr int X6_PATCH_BYTES = 3 * X6_NPOS_PAD * 16;      // 9984
constexpr int X6_W_BYTES = 9 * X6_COT * 3 * 16;           // 27648 = 27 x 1024
constexpr int X6_PIECES = X6_W_BYTES / 1024;

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_value(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32: bits 15:0 = bf16(a), bits 31:16 = bf16(b), round to nearest even
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// x -> (hi, mid, lo) bf16 bit patterns with x == hi + mid + lo (exactly, barring underflow of the last part)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_bits(x);
    const float r1 = x - bf16_value(h);
    m = bf16_bits(r1);
    const float r2 = r1 - bf16_value(m);
    l = bf16_bits(r2);
}

// bank[dir][chunk][cotile][tap][co][part][ch]: fwd: co = output channel, ch = input channel, tap = ky*3+kx;
// bwd-data: roles swapped and taps flipped.  Zero padding for channels beyond the tensor.
__global__ void pack_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int cout, int cin,
                               int backward) {
    const int CO = backward ? cin : cout;   // channels produced by the pass
    const int CI = backward ? cout : cin;   // channels consumed
    const int nchunk = (CI + 7) / 8, ntile = (CO + X6_COT - 1) / X6_COT;
    const int64_t total = (int64_t)nchunk * ntile * 9 * X6_COT * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int ch = (int)(r % 8);
        r /= 8;
        const int co = (int)(r % X6_COT);
        r /= X6_COT;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * X6_COT + co, i = chunk * 8 + ch;
        float v = 0.f;
        if (o < CO && i < CI) {
            if (!backward) v = w[((int64_t)o * cin + i) * 9 + tap];
            else v = w[((int64_t)i * cin + o) * 9 + (8 - tap)];
        }
        unsigned short h, m, l;
        split3(v, h, m, l);
        const int64_t base = ((((int64_t)chunk * ntile + tile) * 9 + tap) * X6_COT + co) * 24 + ch;
        bank[base] = h;
        bank[base + 8] = m;
        bank[base + 16] = l;
    }
}

template <bool TL, bool ACC, bool OM, bool C8>
__global__ void __launch_bounds__(256, 4) conv_x6_kernel(ConvArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[X6_PATCH_BYTES + X6_W_BYTES];
    unsigned char* Pl = smem;                    // [part][pos][16 B]
    unsigned char* Wl = smem + X6_PATCH_BYTES;   // [tap][co][part][16 B]

#ifdef MAUA_X6_STAMP
    const unsigned long long st_enter = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    const int cotile = blockIdx.y;
    const int co0 = cotile * X6_COT;
    const int ntile = gridDim.y;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    // XCD-aware tile order (and optionally persistent workgroups): workgroups are dealt to the 8 XCDs round-robin by linear id, so
    // XCD k owns the k-th contiguous band of tiles (row-major; halo rows / columns shared by neighbouring tiles are hits
    // in ITS L2) and workgroup `slot` of that XCD walks the band with stride `slots`.  The K loop below runs over the
    // flattened (tile, chunk) sequence: the first chunk of the next tile is prefetched behind the last chunk of the
    // current one, so prologue, epilogue and workgroup turn-around are paid once per workgroup, not once per tile
    // (measured before: 17 us per 38 us of K loop on conv1_2).
    const int tiles_total = p.tiles_x * ((p.OH + X6_PH - 1) / X6_PH);
    const int per_xcd = (tiles_total + 7) >> 3;
    const int slots = gridDim.x >> 3;  // grid.x is a multiple of 8
    const int band_hi = min(((int)(blockIdx.x & 7) + 1) * per_xcd, tiles_total);
    int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= band_hi) return;  // whole workgroup leaves: no barrier is skipped

    // staging descriptor of this thread's patch position for the tile being STAGED (the current tile, or the next one while
    // its first chunk is prefetched): a 32-bit byte offset inside a channel plane (clamped into the image; positions in the
    // zero padding are blanked when the chunk is written to LDS)
    unsigned p_byte = 0;
    bool pos_ok = false;
    auto set_stage = [&](int t) {
        const int x0 = (t % p.tiles_x) * 32, y0 = (t / p.tiles_x) * X6_PH;
        const int r = tid / X6_PC, col = tid - r * X6_PC;
        const int iy = y0 + r - p.pad, ix = x0 + col - p.pad;
        pos_ok = tid < X6_NPOS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        p_byte = pos_ok ? (unsigned)(iy * p.W + ix) * 4u : 0u;
    };
    set_stage(tile);
    float rp[8];
    int rp_c0 = 0;  // first channel of the chunk held in rp
    auto load_patch = [&](int c0) {
        asm volatile("" : "+s"(c0));
        // Unconditional loads, wave-uniform plane base + per-lane 32-bit offset (no vector address arithmetic); NOTHING
        // touches the loaded registers here, so the wait for them sits in store_patch (after the chunk's MFMAs).
        rp_c0 = c0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chn = C8 ? c0 + c : min(c0 + c, p.Cin - 1);
            const char* plane = reinterpret_cast<const char*>(xin + (int64_t)chn * in_plane);
            rp[c] = *reinterpret_cast<const float*>(plane + p_byte);
        }
    };
    // The split is pure VALU work on registers: it runs behind the chunk's last MFMAs (which keep the matrix pipe busy
    // for ~400 cycles after they are issued); only the three LDS writes have to wait for the barrier that frees the patch.
    u32x4 vh, vm, vl;
    auto split_patch = [&]() {
        // pairs of values with the packed converts: the three u32 words of a pair come out ready to store
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok0 = pos_ok && (C8 || rp_c0 + 2 * q < p.Cin), ok1 = pos_ok && (C8 || rp_c0 + 2 * q + 1 < p.Cin);
            const float v0 = ok0 ? rp[2 * q] : 0.f, v1 = ok1 ? rp[2 * q + 1] : 0.f;
            const unsigned H = cvt_pk_bf16(v0, v1);
            const float r0 = v0 - __builtin_bit_cast(float, H << 16), r1 = v1 - __builtin_bit_cast(float, H & 0xffff0000u);
            const unsigned M = cvt_pk_bf16(r0, r1);
            const float s0 = r0 - __builtin_bit_cast(float, M << 16), s1 = r1 - __builtin_bit_cast(float, M & 0xffff0000u);
            vh[q] = H;
            vm[q] = M;
            vl[q] = cvt_pk_bf16(s0, s1);
        }
    };
    auto write_patch = [&]() {
        if (tid < X6_NPOS_PAD) {
            *reinterpret_cast<u32x4*>(Pl + (0 * X6_NPOS_PAD + tid) * 16) = vh;
            *reinterpret_cast<u32x4*>(Pl + (1 * X6_NPOS_PAD + tid) * 16) = vm;
            *reinterpret_cast<u32x4*>(Pl + (2 * X6_NPOS_PAD + tid) * 16) = vl;
        }
    };
    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    // Filter slice of a chunk = 27 pieces of 1 KiB in LDS order.  Half A = taps 0-3 (pieces 0-11, read by k-steps 0-1),
    // half B = taps 4-8 (pieces 12-26, read by k-steps 2-4).  Every wave issues exactly 3 (A) or 4 (B) LDS-DMA
    // instructions so that the counted vmcnt waits below are the same for all waves (wave 3 repeats piece 26).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane16 = lane * 16;
    auto dma_half = [&](int ch, bool second) {
        const unsigned char* src = bank + ((int64_t)ch * ntile + cotile) * X6_W_BYTES;
        const int first = second ? 12 : 0, count = second ? 4 : 3, last = second ? 26 : 11;
        for (int i = 0; i < count; ++i) {
            const int q = min(first + wv + 4 * i, last);
            // LDS-DMA issued from inline asm so that hipcc does not drain it with vmcnt(0) at the next ds_read
            // (cdna_hip_programming.md §5.7; M0 = wave-uniform LDS byte address, lane i lands at M0 + 16*i)
            // scalar base + 32-bit lane offset: no per-lane 64-bit address to keep (or spill) across the loop
            const unsigned char* g = src + q * 1024;
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(Wl + q * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(lane16), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "s"(g)
                         : "memory");
        }
    };

    // fragment addresses: tap of this lane half for k-step s < 4 is 2s + half
    int b_byte[5], a_byte[5];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tap = 2 * s + half;
        const int ky = tap / 3, kx = tap - 3 * ky;
        b_byte[s] = ((wave + ky) * X6_PC + j + kx) * 16;
        a_byte[s] = ((tap * X6_COT + j) * 3) * 16;
    }
    // the ninth tap has no partner: it runs alone at K = 8 (v_mfma_f32_32x32x8_bf16_1k), lane half h taking channels
    // 4h..4h+3 of that tap.  Same cycles as a K = 16 step padded with zeros, but 8-byte fragment reads and half the
    // multipliers switching: the kernel is power-bound, measured -2 % time.
    b_byte[4] = ((wave + 2) * X6_PC + j + 2) * 16 + half * 8;
    a_byte[4] = ((8 * X6_COT + j) * 3) * 16 + half * 8;

    // two-level accumulation (TL): fold the running sums into a master accumulator (plain fp32 VALU adds, round to
    // nearest) after EVERY chunk, so no MFMA accumulation chain is longer than 5 x 6 instructions.  Measured: folding every
    // chunk halves the pixel-gradient error of folding every fourth (the bf16 MFMA adder truncates toward zero).
    // The accumulators start from the bias (loaded behind the prologue's / the previous epilogue's memory latency) instead
    // of zero, so the epilogue has no dependent loads.  Split-K partial sums start from zero: the finish kernel adds the
    // bias once.
    f32x16 acc[2], master[TL ? 2 : 1];
    auto init_acc = [&]() {
        const bool with_bias = p.bias != nullptr && p.ksplit <= 1;  // wave-uniform
        const float* bias = p.bias;
        int cbase = co0;
        asm volatile("" : "+s"(bias), "+s"(cbase));  // opaque: 32 hoisted per-lane addresses would live across the tile loop
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cbase + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float b0 = 0.f;
                if (with_bias) b0 = bias[min(co, p.Cout - 1)];
                if constexpr (TL) {
                    master[t][r] = b0;
                    acc[t][r] = 0.f;
                } else {
                    acc[t][r] = b0;
                }
            }
    };
    init_acc();

    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto kstep_tap9 = [&]() {
        s16x4 b[3], a[2][3];
#pragma unroll
        for (int part = 0; part < 3; ++part)
            b[part] = *reinterpret_cast<const s16x4*>(Pl + part * X6_NPOS_PAD * 16 + b_byte[4]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 3; ++part)
                a[t][part] = *reinterpret_cast<const s16x4*>(Wl + a_byte[4] + (t * 32 * 3 + part) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][2], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][0], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };
    auto kstep = [&](int s) {
        bf16x8 b[3], a[2][3];
#pragma unroll
        for (int part = 0; part < 3; ++part)
            b[part] = *reinterpret_cast<const bf16x8*>(Pl + part * X6_NPOS_PAD * 16 + b_byte[s]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 3; ++part)
                a[t][part] = *reinterpret_cast<const bf16x8*>(Wl + a_byte[s] + (t * 32 * 3 + part) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // smallest terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][2], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };

    // Pipeline (raw s_barrier + counted vmcnt, cdna_hip_programming.md "Pipelining across barriers"): the filter halves of
    // chunk c+1 stream in behind the k-steps of chunk c that no longer need the LDS region they overwrite.
    //   chunk c:  [patch(c+1) global loads -> regs]  ks0 ks1 | X1 | DMA A(c+1) | ks2 ks3 ks4, split patch(c+1) | X2 | patch(c+1) -> LDS,
    //             DMA B(c+1), wait A(c+1) | X3 | ... next chunk; B(c+1) is waited for just before the next X1.
    // vmcnt is in issue order: before X1 the queue is [B(c) x4 (old), patch loads x8 (young)] -> vmcnt(8) retires B(c);
    // before X3 it is [A(c+1) x3, B(c+1) x4] -> vmcnt(4) retires A(c+1) (the patch loads were consumed before).
    const int nchunks_all = (p.Cin + 7) / 8;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;        // chunks per split
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);      // this workgroup covers chunks [ch_begin, nchunks)
    dma_half(ch_begin, false);  // the filter DMA and the patch loads are in flight together: one memory latency, not two
    dma_half(ch_begin, true);
    load_patch(ch_begin * 8);
    split_patch();
    write_patch();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef MAUA_X6_STAMP  // diagnostic build only (tools/x6_clock.py): shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_w[6] = {0, 0, 0, 0, 0, 0}, st_t;
    int st_chunks = 1;
#define ST_BEGIN() st_t = __builtin_amdgcn_s_memtime()
#define ST_END(i) st_w[i] += __builtin_amdgcn_s_memtime() - st_t
#else
#define ST_BEGIN()
#define ST_END(i)
#endif
    // Issue priority rotates among the workgroups that share a CU (HW_ID.TG_ID tells them apart): with equal priorities
    // the arbiter always prefers the same (oldest) wave of a SIMD, which then finishes ~20 % early and leaves the matrix
    // pipe to fewer and fewer waves; measured on conv4_2 the last wave ended 139 us after the first.
    const int tg_slot = (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4) & 3);  // hwreg(HW_REG_HW_ID, 16, 4)
    // epilogue of one tile: lane holds pixel column j of row y0+wave; register r is output channel (r&3)+8*(r>>2)+4*half of
    // block t.  No LDS access, so it can run while the next tile's first chunk is already staged.
    auto epilogue = [&](int t_done) {
    const int x0 = (t_done % p.tiles_x) * 32, y0 = (t_done / p.tiles_x) * X6_PH;
    // opaque copy: keeps the per-channel address arithmetic inside the epilogue (hoisted out of the tile loop it costs 64
    // live 64-bit values, i.e. spills)
    long long oplane = out_plane;
    asm volatile("" : "+s"(oplane));
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * oplane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * oplane : nullptr;
    const int oy = y0 + wave, ox = x0 + j;
    const bool pvalid = oy < p.OH && ox < p.OW;
    const int64_t opix = (int64_t)oy * p.OW + ox;
    if (p.ksplit > 1) {  // split-K: raw partial sums, finished by x6_splitk_finish_kernel in split order
        float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * oplane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pvalid && co < p.Cout) wsp[(int64_t)co * oplane + opix] = TL ? master[t][r] : acc[t][r];
            }
        return;
    }
    // One exec mask for the whole epilogue (lanes outside the image leave), per-lane base pointers computed once, and
    // wave-uniform channel offsets: a store costs one 64-bit add, no multiplies, no branch.
    if (pvalid) {
        const bool full = co0 + X6_COT <= p.Cout;  // wave-uniform: every channel of the tile exists
        const int64_t lane_off = (int64_t)(co0 + 4 * half) * oplane + opix;
        float* __restrict__ yl = yout + lane_off;
        const float* __restrict__ oml = OM ? om + lane_off : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // all loads of a 16-register block first (previous value, ReLU mask), then the stores: a load issued
            // after a store to the same array would wait for it, turning the epilogue into 32 serial memory round trips
            float prev[16], msk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);  // compile-time channel offset inside the tile
                const int64_t o = (full || co0 + cr + 4 * half < p.Cout) ? (int64_t)cr * oplane : 0;
                // compile-time switches: with run-time flags hipcc branches around every load and waits for each one
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = yl[o];
                if constexpr (OM) msk[r] = oml[o];
            }
            float outv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = TL ? master[t][r] : acc[t][r];
                v += prev[r];
                if (p.relu) v = v > 0.f ? v : 0.f;
                outv[r] = msk[r] > 0.f ? v : 0.f;
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yl[(int64_t)(t * 32 + (r & 3) + 8 * (r >> 2)) * oplane] = outv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                    if (co0 + cr + 4 * half < p.Cout) yl[(int64_t)cr * oplane] = outv[r];
                }
            }
        }
    }
    };
    int ch = ch_begin;
    for (;;) {
        const bool last = ch + 1 >= nchunks;          // last chunk of the current tile
        const int ntile = tile + slots;               // next tile of this workgroup
        const bool more = !last || ntile < band_hi;   // there is a chunk to prefetch
        const int nch = last ? ch_begin : ch + 1;     // ... this one (of `ntile` when `last`)
        switch ((tg_slot + ch) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        if (more) {
            if (last) set_stage(ntile);
            load_patch(nch * 8);
        }
        kstep(0);
        kstep(1);
        // B(ch) must have landed (all waves) before k-step 2; A(ch)'s readers are done after this barrier.  vmcnt(8) leaves
        // only the 8 patch loads above in flight: everything older (B(ch), a previous tile's epilogue stores) is waited for.
        ST_BEGIN();
        if (more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ST_END(0);
        ST_BEGIN();
        __builtin_amdgcn_s_barrier();
        ST_END(1);
        if (more) dma_half(nch, false);
        kstep(2);
        kstep(3);
        kstep_tap9();
        if (more) split_patch();
        if constexpr (TL) {  // fold every chunk
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    master[t][r] += acc[t][r];
                    acc[t][r] = 0.f;
                }
        }
        ST_BEGIN();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // X2: every wave is done reading the patch and half B
        ST_END(2);
        if (more) {
            ST_BEGIN();
            write_patch();
            dma_half(nch, true);
            ST_END(3);
            ST_BEGIN();
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");  // A(next) landed, patch writes done
            ST_END(4);
            ST_BEGIN();
            __builtin_amdgcn_s_barrier();  // X3
            ST_END(5);
        }
        if (last) {
            epilogue(tile);
            if (!more) break;
            tile = ntile;
            init_acc();
        }
        ch = nch;
#ifdef MAUA_X6_STAMP
        ++st_chunks;
#endif
    }

#ifdef MAUA_X6_STAMP
    if (p.ws && p.ksplit <= 1 && lane == 0) {  // the stamps go to a buffer nothing else reads
        const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* st = reinterpret_cast<unsigned long long*>(p.ws) +
                                 2 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 4 + blockIdx.x * 4 + wave);
        st[0] = st_c1 - st_c0;
        st[1] = st_r1 - st_r0;
        st[2 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4] = st_r0;      // absolute 100 MHz stamps: second / third planes
        st[4 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4] = st_r1;
        st[2 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4 + 1] = st_enter;
        for (int i = 0; i < 6; ++i) st[2 * (4 + i) * (size_t)gridDim.x * gridDim.y * gridDim.z * 4] = st_w[i];
        st[6 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4 + 1] = (unsigned long long)st_chunks;
    }
#endif
#ifdef MAUA_X6_STAMP
    if (p.ws && p.ksplit <= 1 && lane == 0) {
        const unsigned long long st_issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* st = reinterpret_cast<unsigned long long*>(p.ws) +
                                 2 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 4 + blockIdx.x * 4 + wave);
        st[4 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4 + 1] = st_issued;
        st[6 * (size_t)gridDim.x * gridDim.y * gridDim.z * 4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// Split the K loop over several workgroups when the output grid quantises badly against the chip's 1024 workgroup slots
// (4 per CU x 256 CUs): e.g. conv4_x of a 724-px image is 552 workgroups - 54 % of one round - and of a 1448-px image
// 2208 = 2.16 rounds.  Cost model per candidate split: rounds x (chunks per workgroup + 1.5 chunk-times of prologue /
// epilogue) x 5 us per chunk, plus the finish kernel's traffic (ks partial reads + one write at ~5 TB/s).
static int x6_choose_split(const ConvArgs& a, int n) {
    const int64_t wgs = (int64_t)((a.OW + 31) / 32) * ((a.OH + X6_PH - 1) / X6_PH) * ((a.Cout + X6_COT - 1) / X6_COT) * split_batch_hint();  // planned frames (see conv_x3w.hip)
    const int nchunks = (a.Cin + 7) / 8;
    if (wgs >= 4096 || nchunks < 8) return 1;
    const double out_mb = (double)split_batch_hint() * a.Cout * a.OH * a.OW * 4.0 / 1e6;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 16 && ks <= nchunks / 4; ++ks) {
        const double rounds = (double)((wgs * ks + 1023) / 1024);
        double cost = rounds * ((double)((nchunks + ks - 1) / ks) + 1.5) * 5.0;
        if (ks > 1) cost += (ks + 1) * out_mb / 5.0 + 5.0;  // finish kernel: MB / (5 TB/s) in us, plus its launch
        if (cost < best_cost * 0.97) {  // prefer the smaller split unless the gain is real
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

int conv_x6_launch(const ConvArgs& a, int n, hipStream_t stream) {
    ConvArgs p = a;
    p.tiles_x = (a.OW + 31) / 32;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + X6_PH - 1) / X6_PH);
    int ks = a.ws ? x6_choose_split(a, n) : 1;
#ifdef MAUA_X6_STAMP
    ks = 1;  // the workspace is the stamp buffer in the diagnostic build
#endif
    p.ksplit = ks;
    // One workgroup per tile by default (the hardware's dynamic scheduling copes best with grids that do not divide the
    // 1024 workgroup slots evenly).  MAUA_X6_PERSIST=1 launches persistent workgroups instead - enough to fill the chip
    // once, each walking several tiles of its XCD band with cross-tile prefetch; measured equal on 1024x1024 (the other
    // workgroups of a CU already hide a workgroup's prologue / epilogue), and it loses on uneven tile counts.
    const int64_t cot = (a.Cout + X6_COT - 1) / X6_COT, per_xcd = (tiles + 7) / 8;
    const bool persist = tuning("x6_persist", 0) != 0;
    int64_t g8 = per_xcd;
    if (persist) {
        g8 = (1024 + cot * n * ks * 8 - 1) / (cot * n * ks * 8);
        g8 = g8 < 1 ? 1 : (g8 > per_xcd ? per_xcd : g8);
    }
    dim3 grid((unsigned)(g8 * 8), (unsigned)cot, (unsigned)(n * ks));
    const bool tl = (a.Cin + 7) / 8 > 4, acc = ks == 1 && a.accumulate != 0, om = ks == 1 && a.omask != nullptr;
#define MAUA_X6_LAUNCH(TL_, ACC_, OM_)                                                                      \
    do {                                                                                                    \
        if (a.Cin % 8 == 0) hipLaunchKernelGGL((conv_x6_kernel<TL_, ACC_, OM_, true>), grid, dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((conv_x6_kernel<TL_, ACC_, OM_, false>), grid, dim3(256), 0, stream, p);                \
    } while (0)
    if (tl) {
        if (acc && om) MAUA_X6_LAUNCH(true, true, true);
        else if (acc) MAUA_X6_LAUNCH(true, true, false);
        else if (om) MAUA_X6_LAUNCH(true, false, true);
        else MAUA_X6_LAUNCH(true, false, false);
    } else {
        if (acc && om) MAUA_X6_LAUNCH(false, true, true);
        else if (acc) MAUA_X6_LAUNCH(false, true, false);
        else if (om) MAUA_X6_LAUNCH(false, false, true);
        else MAUA_X6_LAUNCH(false, false, false);
    }
#undef MAUA_X6_LAUNCH
    int rc = check_launch("conv_x6_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_x6_bank_bytes(int cout_produced, int cin_consumed) {
    if (cout_produced <= 0 || cin_consumed <= 0 || cout_produced > (1 << 20) || cin_consumed > (1 << 20)) return 0;
    const size_t nchunk = (cin_consumed + 7) / 8, ntile = (cout_produced + X6_COT - 1) / X6_COT;
    return nchunk * ntile * X6_W_BYTES;
}

int maua_conv_pack_filters_x6(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20), MAUA_E_INVAL, "conv_pack_filters_x6: bad args");
    if (bank_fwd) {
        hipLaunchKernelGGL(pack_x6_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_fwd,
                           cout, cin, 0);
        int rc = check_launch("pack_x6_kernel");
        if (rc) return rc;
    }
    if (bank_bwd) {
        hipLaunchKernelGGL(pack_x6_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)bank_bwd,
                           cout, cin, 1);
        return check_launch("pack_x6_kernel");
    }
    return MAUA_OK;
}

size_t maua_conv_x6_workspace_bytes(int n, int cin, int h, int w, int cout, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad)) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - 2;
    a.OW = w + 2 * pad - 2;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const int ks = x6_choose_split(a, n);
    return ks > 1 ? (size_t)n * ks * cout * a.OH * a.OW * sizeof(float) : 0;
}

int maua_conv3x3_x6(const float* x, const void* bank, const float* bias, const float* out_relu_mask, float* y, int n,
                    int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y, MAUA_E_INVAL, "conv3x3_x6: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && pad <= 2, MAUA_E_INVAL, "conv3x3_x6: bad dims");
    MAUA_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, MAUA_E_UNSUPPORTED, "conv3x3_x6: input smaller than the filter");
    MAUA_REQUIRE((int64_t)h * w < (1ll << 30), MAUA_E_UNSUPPORTED, "conv3x3_x6: plane too large");
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
    // split-K only with a big enough workspace; without one the kernel still runs, just with fewer workgroups
    a.ws = (workspace && workspace_bytes >= maua_conv_x6_workspace_bytes(n, cin, h, w, cout, pad)) ? (float*)workspace : nullptr;
#ifdef MAUA_X6_STAMP
    a.ws = (float*)workspace;  // caller provides 64 B per workgroup
#endif
    return conv_x6_launch(a, n, (hipStream_t)stream);
}

}  // extern "C"
