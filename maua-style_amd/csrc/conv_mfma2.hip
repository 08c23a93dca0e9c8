// Stride-1 implicit-GEMM convolution on the fp32 matrix cores (second generation: the first one reached 61 % / 42 %
// MFMA utilisation forward / backward; this is the restructuring its profile suggested).
//
//   * channel-interleaved LDS images: patch[pos][8] and filters[tap][co][8], where the 8 floats of a position are
//     the chunk's 8 input channels ordered (half, cp) with channel = 2*cp + half.  One ds_read_b128 then feeds the
//     A (or B) operand of FOUR consecutive MFMAs (k-pairs cp = 0..3), so a tap needs 4 LDS reads for 16 MFMAs
//     instead of 16 reads, and the reads of tap t+1 are issued before the MFMAs of tap t (software pipeline);
//   * the two 16-byte slots of a position are swapped when bit2^bit3 of the position is set: with that swizzle the
//     b128 reads of every 16-lane group and the b128 staging writes of every aligned 8-lane group are bank-conflict
//     free (MI355X_MICROARCH.md §LDS: b128 reads bank = (addr/4) % 64 over groups {0-3,12-15,20-27}...);
//   * branch-free staging: out-of-image / out-of-range elements load from a clamped address and are zeroed by a
//     select, so the 12 (+12 mask) + 20 global loads of a chunk issue back to back;
//   * KC = 8 input channels per chunk for every layer (conv1_1's 3 channels are zero-padded inside LDS only).
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>

#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int slot_swz(int p) { return ((p >> 2) ^ (p >> 3)) & 1; }

template <int KS, int TCO, int TPX, bool TL, bool MASK, bool ACC, bool OM>
__global__ void __launch_bounds__(256, 2) conv_mfma2_kernel(ConvArgs p) {
    constexpr int KC = 8;
    constexpr int CO_T = 32 * TCO;
    constexpr int PH = 4 * TPX;
    constexpr int PR = PH + KS - 1, PC = 32 + KS - 1;
    constexpr int NPOS = PR * PC;
    constexpr int NPOS_PAD = (NPOS + 7) / 8 * 8;
    constexpr int PATCH = NPOS_PAD * KC;        // floats
    constexpr int WCH = KS * KS * CO_T * KC;    // floats
    constexpr int BUF = PATCH + WCH;
    constexpr int NPI = (NPOS_PAD * 2 + 255) / 256;        // patch staging items per thread (pos, half)
    constexpr int NWI = (KS * KS * CO_T * 2 + 255) / 256;  // filter staging items per thread (tap, co, half)
    constexpr int FLUSH = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;  // split-K: blockIdx.z = n * ksplit + split (see conv_x6.hip)
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.z / ksplit);  // the division runs on the vector ALU
    const int split = blockIdx.z - n * ksplit;
    const int co0 = blockIdx.y * CO_T;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    const float* __restrict__ xmask = MASK ? p.mask + (int64_t)n * p.Cin * in_plane : nullptr;

    int x0 = 0, y0 = 0;
    int64_t lin0 = 0;
    if constexpr (KS == 1) {
        lin0 = (int64_t)blockIdx.x * (PH * 32);
    } else {
        x0 = (blockIdx.x % p.tiles_x) * 32;
        y0 = (blockIdx.x / p.tiles_x) * PH;
    }

    // ---- staging descriptors (chunk-invariant; LDS slots are recomputed from the item id when stored) --------------
    int p_off[NPI];  // offset inside an input plane, -1 = zero padding / unused item
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int q = tid + 256 * i;
        const int h = q / NPOS_PAD, pos = q - h * NPOS_PAD;
        p_off[i] = -1;
        if (h < 2 && pos < NPOS) {
            const int r = pos / PC, col = pos - r * PC;
            if constexpr (KS == 1) {
                const int64_t pix = lin0 + r * 32 + col;
                if (pix < in_plane) p_off[i] = (int)pix;
            } else {
                const int iy = y0 + r - p.pad, ix = x0 + col - p.pad;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) p_off[i] = iy * p.W + ix;
            }
        }
    }
    int w_off[NWI];  // (tap*Cin + h)*Cout + co0 + co, -1 = column out of range / unused item
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int q = tid + 256 * i;
        const int co = q % CO_T, h = (q / CO_T) & 1, tap = q / (2 * CO_T);
        w_off[i] = -1;
        if (tap < KS * KS && co0 + co < p.Cout) w_off[i] = (tap * p.Cin + h) * p.Cout + co0 + co;
    }

    f32x4 rp[NPI], rw[NWI];
    auto load_patch = [&](int c0) {
        c0 = __builtin_amdgcn_readfirstlane(c0);  // opaque + scalar: addresses stay a function of c0, no per-load running pointers
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int h = (tid + 256 * i) / NPOS_PAD;
            const bool sp_ok = p_off[i] >= 0;
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                const int c = c0 + 2 * cp + h;
                const bool ok = sp_ok && c < p.Cin;
                const int64_t a = ok ? (int64_t)c * in_plane + p_off[i] : 0;
                float v = xin[a];
                if constexpr (MASK) v = xmask[a] > 0.f ? v : 0.f;
                rp[i][cp] = ok ? v : 0.f;
            }
        }
    };
    constexpr int NWA = (NWI + 1) / 2;  // filter items staged in the first of two phases
    auto load_filters = [&](int c0, int lo, int hi) {
        c0 = __builtin_amdgcn_readfirstlane(c0);
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            if (i < lo || i >= hi) continue;
            const int h = ((tid + 256 * i) / CO_T) & 1;
            const bool co_ok = w_off[i] >= 0;
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                const bool ok = co_ok && c0 + 2 * cp + h < p.Cin;
                const int64_t a = ok ? (int64_t)w_off[i] + (int64_t)(c0 + 2 * cp) * p.Cout : 0;
                const float v = p.w[a];
                rw[i][cp] = ok ? v : 0.f;
            }
        }
    };
    auto store_patch = [&](int buf) {
        float* xl = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int q = tid + 256 * i;
            const int h = q / NPOS_PAD, pos = q - h * NPOS_PAD;
            if (h < 2) *reinterpret_cast<f32x4*>(xl + pos * KC + 4 * (h ^ slot_swz(pos))) = rp[i];
        }
    };
    auto store_filters = [&](int buf, int lo, int hi) {
        float* wl = smem + buf * BUF + PATCH;
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            if (i < lo || i >= hi) continue;
            const int q = tid + 256 * i;
            const int co = q % CO_T, h = (q / CO_T) & 1, tap = q / (2 * CO_T);
            if (tap < KS * KS) *reinterpret_cast<f32x4*>(wl + (tap * CO_T + co) * KC + 4 * (h ^ slot_swz(co))) = rw[i];
        }
    };

    // ---- fragment read offsets (floats) ------------------------------------------------------------------------
    int a_off[TCO];
#pragma unroll
    for (int t = 0; t < TCO; ++t) {
        const int co = t * 32 + j;
        a_off[t] = co * KC + 4 * (half ^ slot_swz(co));
    }
    int b_pos[TPX];
#pragma unroll
    for (int u = 0; u < TPX; ++u) b_pos[u] = (wave * TPX + u) * PC + j;
    auto b_off = [&](int u, int tap) {
        const int pos = b_pos[u] + (tap / KS) * PC + (tap % KS);
        return pos * KC + 4 * (half ^ slot_swz(pos));
    };

    f32x16 acc[TCO][TPX];
    f32x16 master[TL ? TCO : 1][TL ? TPX : 1];
#pragma unroll
    for (int t = 0; t < TCO; ++t)
#pragma unroll
        for (int u = 0; u < TPX; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[t][u][r] = 0.f;
                if constexpr (TL) master[t][u][r] = 0.f;
            }

    // Staging of chunk ch+1 is spread over chunk ch's taps so that few staging registers are live at once: patch loads
    // at tap 0; at tap P1 patch -> LDS and the first half of the filter loads; at tap P2 those -> LDS and the second half
    // of the filter loads; the rest -> LDS after the last tap.  The other LDS buffer is free for the whole chunk (its
    // readers finished before the barrier).
    constexpr int P1 = KS == 1 ? 99 : (KS * KS) / 3, P2 = KS == 1 ? 99 : (2 * KS * KS) / 3;
    const int nchunks_all = (p.Cin + KC - 1) / KC;
    const int cps = __builtin_amdgcn_readfirstlane((nchunks_all + ksplit - 1) / ksplit);
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);  // this workgroup covers chunks [ch_begin, nchunks)
    load_patch(ch_begin * KC);
    store_patch(ch_begin & 1);
    load_filters(ch_begin * KC, 0, NWI);
    store_filters(ch_begin & 1, 0, NWI);
    __syncthreads();
    for (int ch = ch_begin; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < nchunks;
        const float* xl = smem + cur * BUF;
        const float* wl = xl + PATCH;

        f32x4 fa[2][TCO], fb[2][TPX];
#pragma unroll
        for (int t = 0; t < TCO; ++t) fa[0][t] = *reinterpret_cast<const f32x4*>(wl + a_off[t]);
#pragma unroll
        for (int u = 0; u < TPX; ++u) fb[0][u] = *reinterpret_cast<const f32x4*>(xl + b_off(u, 0));
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int cb = tap & 1, nb = cb ^ 1;
            if (tap == 0 && more) load_patch((ch + 1) * KC);
            if (tap == P1 && more) {
                store_patch(cur ^ 1);
                load_filters((ch + 1) * KC, 0, NWA);
            }
            if (tap == P2 && more) {
                store_filters(cur ^ 1, 0, NWA);
                load_filters((ch + 1) * KC, NWA, NWI);
            }
            if (tap + 1 < KS * KS) {
#pragma unroll
                for (int t = 0; t < TCO; ++t)
                    fa[nb][t] = *reinterpret_cast<const f32x4*>(wl + (tap + 1) * CO_T * KC + a_off[t]);
#pragma unroll
                for (int u = 0; u < TPX; ++u) fb[nb][u] = *reinterpret_cast<const f32x4*>(xl + b_off(u, tap + 1));
            }
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                for (int t = 0; t < TCO; ++t)
#pragma unroll
                    for (int u = 0; u < TPX; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][t][cp], fb[cb][u][cp], acc[t][u], 0, 0, 0);
        }
        if constexpr (TL) {
            if (((ch - ch_begin) & (FLUSH - 1)) == FLUSH - 1 || ch + 1 == nchunks) {
#pragma unroll
                for (int t = 0; t < TCO; ++t)
#pragma unroll
                    for (int u = 0; u < TPX; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            master[t][u][r] += acc[t][u][r];
                            acc[t][u][r] = 0.f;
                        }
            }
        }
        if (more) {
            if constexpr (KS == 1) {  // a single tap: nothing to hide the staging behind, do it after the MFMAs
                store_patch(cur ^ 1);
                load_filters((ch + 1) * KC, 0, NWI);
                store_filters(cur ^ 1, 0, NWI);
            } else {
                store_filters(cur ^ 1, NWA, NWI);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
    const float* __restrict__ om = p.omask ? p.omask + (int64_t)n * p.Cout * out_plane : nullptr;
#pragma unroll
    for (int u = 0; u < TPX; ++u) {
        int64_t opix;
        bool pvalid;
        if constexpr (KS == 1) {
            opix = lin0 + (wave * TPX + u) * 32 + j;
            pvalid = opix < out_plane;
        } else {
            const int oy = y0 + wave * TPX + u, ox = x0 + j;
            pvalid = oy < p.OH && ox < p.OW;
            opix = (int64_t)oy * p.OW + ox;
        }
        if (p.ksplit > 1) {  // raw partial sums; conv_splitk_finish_kernel adds them in split order
            float* wsp = p.ws + (int64_t)blockIdx.z * p.Cout * out_plane;
#pragma unroll
            for (int t = 0; t < TCO; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pvalid && co < p.Cout) wsp[(int64_t)co * out_plane + opix] = TL ? master[t][u][r] : acc[t][u][r];
                }
            continue;
        }
#pragma unroll
        for (int t = 0; t < TCO; ++t) {
            float prev[16], msk[16];  // loads first, stores second (see conv_x6.hip)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int64_t o = (pvalid && co < p.Cout) ? (int64_t)co * out_plane + opix : 0;
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = yout[o];
                if constexpr (OM) msk[r] = om[o];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pvalid && co < p.Cout) {
                    float v = TL ? master[t][u][r] : acc[t][u][r];
                    if (p.bias) v += p.bias[co];
                    v += prev[r];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    yout[(int64_t)co * out_plane + opix] = msk[r] > 0.f ? v : 0.f;
                }
            }
        }
    }
}

template <int KS, int TCO, int TPX, bool TL, bool MASK, bool ACC, bool OM>
static int launch2e(const ConvArgs& a, int n, hipStream_t stream) {
    constexpr int CO_T = 32 * TCO, PH = 4 * TPX;
    constexpr int PR = PH + KS - 1, PC = 32 + KS - 1;
    constexpr int NPOS_PAD = (PR * PC + 7) / 8 * 8;
    constexpr size_t lds = 2ull * (NPOS_PAD * 8 + KS * KS * CO_T * 8) * sizeof(float);
    ConvArgs p = a;
    int64_t tiles;
    if (KS == 1) {
        tiles = ((int64_t)a.OH * a.OW + PH * 32 - 1) / (PH * 32);
        p.tiles_x = 1;
    } else {
        p.tiles_x = (a.OW + 31) / 32;
        tiles = (int64_t)p.tiles_x * ((a.OH + PH - 1) / PH);
    }
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    dim3 grid((unsigned)tiles, (unsigned)((a.Cout + CO_T - 1) / CO_T), (unsigned)(n * ks));
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma2_kernel<KS, TCO, TPX, TL, MASK, ACC, OM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_mfma2_kernel<KS, TCO, TPX, TL, MASK, ACC, OM>), grid, dim3(256), lds, stream, p);
    int rc = check_launch("conv_mfma2_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(a, n, ks, stream);
}

template <int KS, int TCO, int TPX, bool TL, bool MASK>
static int launch2(const ConvArgs& a, int n, hipStream_t stream) {
    const bool split = a.ksplit > 1;  // the finish kernel applies accumulate / mask / bias / ReLU
    const bool acc = !split && a.accumulate != 0, om = !split && a.omask != nullptr;
    if (acc && om) return launch2e<KS, TCO, TPX, TL, MASK, true, true>(a, n, stream);
    if (acc) return launch2e<KS, TCO, TPX, TL, MASK, true, false>(a, n, stream);
    if (om) return launch2e<KS, TCO, TPX, TL, MASK, false, true>(a, n, stream);
    return launch2e<KS, TCO, TPX, TL, MASK, false, false>(a, n, stream);
}

// Round 4: this fp32-MFMA route is the general entry point's arithmetic (maua_conv2d_fwd / _bwd_data) and the A/B reference of the
// split-precision kernels (MAUA_CONV_X6=0), not the product's hot path: one tile shape per filter size and two-level accumulation
// always (24 kernel instances instead of 112: the library shrinks by 2 MB; the single-level and small-grid variants bought 5-10 % on
// geometries nothing dispatches here any more).
template <int KS, int TCO, int TPX>
static int launch2_variant(const ConvArgs& a, int n, hipStream_t stream) {
    if (a.mask) return launch2<KS, TCO, TPX, true, true>(a, n, stream);
    return launch2<KS, TCO, TPX, true, false>(a, n, stream);
}

// Split the channel loop when the output grid is far too small for 256 CUs (deep 1x1 layers on 31x31 maps ...).
int conv_mfma2_choose_split(const ConvArgs& a, int ks, int n) {
    if (ks != 1 && ks != 3 && ks != 5) return 1;
    const int64_t opix = (int64_t)a.OH * a.OW;
    const int64_t tiles = (ks == 1) ? (opix + 127) / 128 : (int64_t)((a.OW + 31) / 32) * ((a.OH + 3) / 4);
    const int64_t wgs = tiles * ((a.Cout + 63) / 64) * split_batch_hint();  // planned frames (see conv_x3w.hip)
    const int nchunks = (a.Cin + 7) / 8;
    if (wgs >= 384 || nchunks < 16) return 1;
    int s = (int)((768 + wgs - 1) / wgs);
    if (s > nchunks / 8) s = nchunks / 8;  // at least 8 chunks per workgroup
    return s < 2 ? 1 : s;
}

int conv_mfma_dispatch(const ConvArgs& a0, int ks, int n, hipStream_t stream) {
    ConvArgs a = a0;
    a.ksplit = a.ws ? conv_mfma2_choose_split(a, ks, n) : 1;
    switch (ks) {
        case 1:
            return launch2_variant<1, 2, 1>(a, n, stream);
        case 3:
            return launch2_variant<3, 2, 1>(a, n, stream);
        case 5:  // 25 taps x 64 channels of filters would leave one workgroup per CU: 32-channel tiles keep two
            return launch2_variant<5, 1, 2>(a, n, stream);
        default:
            set_error("conv_mfma2: kernel size %d not instantiated", ks);
            return MAUA_E_UNSUPPORTED;
    }
}

}  // namespace maua
