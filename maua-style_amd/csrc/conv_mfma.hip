// Stride-1 convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
//   D[co][pixel] = sum_{tap, ci} Wt[tap][ci][co] * X[ci][pixel shifted by tap]
//
// v_mfma_f32_32x32x2_f32: A = filter bank (row i = output channel), B = input patch (column j = pixel), so each
// accumulator register of a lane holds one output channel for the lane's pixel and every store instruction
// writes 32 consecutive pixels (128 B) of two channels - NCHW stays coalesced without a transpose.
//
// One workgroup = 4 waves computes CO_T = 32*TCO output channels x (4*TPX rows x 32 columns) output pixels.
// Per chunk of KC input channels the patch (with its KS-1 halo) and the KS*KS*KC*CO_T filter slice are staged
// in LDS (double buffered; global loads of chunk c+1 are in flight while chunk c is multiplied).
// KS == 1 indexes pixels linearly (no halo), which also serves NIN's 1x1 convs and the Gram backward
// gf += D * F (a 1x1 "conv" whose filter bank is the symmetric matrix D).
//
// The same kernel is the backward-data pass: the caller hands the flipped/transposed bank and, optionally, the
// saved ReLU output as `mask` so that threshold_backward is applied while the gradient patch is staged.
#include <stdlib.h>

#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));


template <int KS, int KC, int TCO, int TPX, bool TL>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs p) {
    constexpr int CO_T = 32 * TCO;
    constexpr int PH = 4 * TPX;
    constexpr int PR = PH + KS - 1, PC = 32 + KS - 1;
    constexpr int PATCH = KC * PR * PC;
    constexpr int WCH = KS * KS * KC * CO_T;
    constexpr int NP = (PATCH + 255) / 256;
    constexpr int NW = (WCH + 255) / 256;
    constexpr int BUF = PATCH + WCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int n = blockIdx.z;
    const int co0 = blockIdx.y * CO_T;
    const int64_t in_plane = (int64_t)p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    const float* __restrict__ xmask = p.mask ? p.mask + (int64_t)n * p.Cin * in_plane : nullptr;

    int x0 = 0, y0 = 0;
    int64_t lin0 = 0;
    if constexpr (KS == 1) {
        lin0 = (int64_t)blockIdx.x * (PH * 32);
    } else {
        x0 = (blockIdx.x % p.tiles_x) * 32;
        y0 = (blockIdx.x / p.tiles_x) * PH;
    }

    // per-thread staging slots: spatial offset inside one input plane (-1 = zero padding) and local channel
    int soff[NP];
    int scl[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (PR * PC);
        const int rem = e - c * (PR * PC);
        const int r = rem / PC, col = rem - r * PC;
        scl[i] = c;
        soff[i] = -1;
        if (e < PATCH) {
            if constexpr (KS == 1) {
                const int64_t pix = lin0 + r * 32 + col;
                if (pix < in_plane) soff[i] = (int)pix;
            } else {
                const int iy = y0 + r - p.pad, ix = x0 + col - p.pad;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) soff[i] = iy * p.W + ix;
            }
        }
    }

    float rp[NP], rw[NW];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float v = 0.f;
            const int c = c0 + scl[i];
            if (soff[i] >= 0 && c < p.Cin) {
                const int64_t a = (int64_t)c * in_plane + soff[i];
                v = xin[a];
                if (xmask) v = xmask[a] > 0.f ? v : 0.f;
            }
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = tid + 256 * i;
            float v = 0.f;
            if (e < WCH) {
                const int tap = e / (KC * CO_T);
                const int rem = e - tap * (KC * CO_T);
                const int kc = rem / CO_T, co = rem - kc * CO_T;
                if (c0 + kc < p.Cin && co0 + co < p.Cout)
                    v = p.w[((int64_t)tap * p.Cin + c0 + kc) * p.Cout + co0 + co];
            }
            rw[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* xl = smem + buf * BUF;
        float* wl = xl + PATCH;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + 256 * i;
            if (e < PATCH) xl[e] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = tid + 256 * i;
            if (e < WCH) wl[e] = rw[i];
        }
    };

    // Two-level accumulation (TL): the MFMA chain is a sequential fp32 sum over K = KS*KS*Cin products; folding it
    // into a master accumulator every FLUSH chunks keeps each chain short (<= FLUSH*KC*KS*KS terms), which brings the
    // rounding noise down to that of a blocked CPU GEMM.  It costs TCO*TPX*16 v_add per FLUSH chunks (< 1 %).
    constexpr int FLUSH = 4;
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

    const int nchunks = (p.Cin + KC - 1) / KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
        const float* xl = smem + cur * BUF;
        const float* wl = xl + PATCH;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int cp = 0; cp < KC / 2; ++cp) {
                    const int kc = 2 * cp + half;
                    float a[TCO], b[TPX];
#pragma unroll
                    for (int t = 0; t < TCO; ++t) a[t] = wl[((ky * KS + kx) * KC + kc) * CO_T + t * 32 + j];
#pragma unroll
                    for (int u = 0; u < TPX; ++u) b[u] = xl[kc * (PR * PC) + (wave * TPX + u + ky) * PC + j + kx];
#pragma unroll
                    for (int t = 0; t < TCO; ++t)
#pragma unroll
                        for (int u = 0; u < TPX; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
                }
            }
        }
        if constexpr (TL) {
            if ((ch & (FLUSH - 1)) == FLUSH - 1 || ch + 1 == nchunks) {
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
        if (ch + 1 < nchunks) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds pixel column j; register r is output channel (r&3) + 8*(r>>2) + 4*half of its 32-block
    float* __restrict__ yout = p.y + (int64_t)n * p.Cout * out_plane;
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
#pragma unroll
        for (int t = 0; t < TCO; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pvalid && co < p.Cout) {
                    float v = TL ? master[t][u][r] : acc[t][u][r];
                    if (p.bias) v += p.bias[co];
                    const int64_t o = (int64_t)co * out_plane + opix;
                    if (p.accumulate) v += yout[o];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    if (p.omask) v = p.omask[(int64_t)n * p.Cout * out_plane + o] > 0.f ? v : 0.f;
                    yout[o] = v;
                }
            }
        }
    }
}

template <int KS, int KC, int TCO, int TPX, bool TL>
static int launch_tl(const ConvArgs& a, int n, hipStream_t stream) {
    constexpr int CO_T = 32 * TCO, PH = 4 * TPX;
    constexpr int PR = PH + KS - 1, PC = 32 + KS - 1;
    constexpr size_t lds = 2ull * (KC * PR * PC + KS * KS * KC * CO_T) * sizeof(float);
    ConvArgs p = a;
    int64_t tiles;
    if (KS == 1) {
        tiles = ((int64_t)a.OH * a.OW + PH * 32 - 1) / (PH * 32);
        p.tiles_x = 1;
    } else {
        p.tiles_x = (a.OW + 31) / 32;
        tiles = (int64_t)p.tiles_x * ((a.OH + PH - 1) / PH);
    }
    dim3 grid((unsigned)tiles, (unsigned)((a.Cout + CO_T - 1) / CO_T), (unsigned)n);
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<KS, KC, TCO, TPX, TL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<KS, KC, TCO, TPX, TL>), grid, dim3(256), lds, stream, p);
    return check_launch("conv_mfma_kernel");
}

template <int KS, int KC, int TCO, int TPX>
static int launch_variant(const ConvArgs& a, int n, hipStream_t stream) {
    // two-level accumulation only where the K loop is long enough to need it
    if ((a.Cin + KC - 1) / KC > 4) return launch_tl<KS, KC, TCO, TPX, true>(a, n, stream);
    return launch_tl<KS, KC, TCO, TPX, false>(a, n, stream);
}

// Picks the tile variant: the 8-row tile unless that leaves fewer than two workgroups per CU.
int conv_mfma_dispatch(const ConvArgs& a, int ks, int n, hipStream_t stream) {
    static const bool use_v1 = getenv("MAUA_CONV_V1") != nullptr;  // A/B switch: first-generation kernel
    if (!use_v1) return conv_mfma2_dispatch(a, ks, n, stream);
    const int64_t opix = (int64_t)a.OH * a.OW;
    const int64_t co_tiles = (a.Cout + 63) / 64;
    const int64_t big_tiles = (ks == 1) ? (opix + 255) / 256 : (int64_t)((a.OW + 31) / 32) * ((a.OH + 7) / 8);
    const bool small_rows = big_tiles * co_tiles * n < 512;
    const bool narrow_co = a.Cout <= 32;
    switch (ks) {
        case 1:
            if (narrow_co) return launch_variant<1, 8, 1, 2>(a, n, stream);
            return small_rows ? launch_variant<1, 8, 2, 1>(a, n, stream) : launch_variant<1, 8, 2, 2>(a, n, stream);
        case 3:
            if (a.Cin <= 4) return launch_variant<3, 4, 2, 2>(a, n, stream);
            if (narrow_co) return launch_variant<3, 8, 1, 2>(a, n, stream);
            return small_rows ? launch_variant<3, 8, 2, 1>(a, n, stream) : launch_variant<3, 8, 2, 2>(a, n, stream);
        case 5:
            return launch_variant<5, 4, 2, 2>(a, n, stream);
        default:
            set_error("conv_mfma: kernel size %d not instantiated", ks);
            return MAUA_E_UNSUPPORTED;
    }
}

}  // namespace maua
