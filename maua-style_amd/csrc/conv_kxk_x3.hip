// KS x KS stride-1 convolution (KS odd: NIN's 5x5 layer, reference models.py:86; forward and backward-data) at fp32-level
// accuracy on the fp16 matrix cores - the fp16x3 arithmetic of conv_x3.hip (two fp16 parts per operand, products
// hh + hl + lh, per-chunk power-of-two scaling of the activations, fp32 master accumulator) without that kernel's
// 3x3-specific software pipeline: with KS*KS = 25 taps a chunk of 8 input channels already carries 78 MFMAs per wave,
// so the pipeline is the plain one: the next chunk's patch AND filter slice are prefetched into registers while the
// current chunk's products run (two workgroups per CU at 60 KB of LDS leave 256 VGPRs per lane), two barriers per chunk.
// Workgroup = 64 output channels x (4 rows x 32 columns), wave = one output row, lane = one column; a K = 16 MFMA step takes
// two taps x 8 channels (lane half h takes tap 2 s + h), the odd last tap runs alone as a K = 8 step.
// LDS: patch [part][(4 + KS - 1) x (32 + KS - 1) positions][8 ch] + filters [tap][part][co][8 ch].
// hipcc-flags: -Xclang -target-feature -Xclang -packed-fp32-ops
#include "common.hpp"

namespace maua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int KX_COT = 64, KX_TR = 4, KX_TC = 32;

// bank[dir][chunk][cotile][tap][part][co][ch] (fp16, pre-scaled by w_scale): fwd: co = output channel, ch = input channel,
// tap = ky*KS+kx; bwd-data: roles swapped and taps flipped.  Zero padding for channels beyond the tensor.
__global__ void pack_kxk_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank, int cout, int cin, int ks,
                                   int backward, float w_scale) {
    const int CO = backward ? cin : cout;
    const int CI = backward ? cout : cin;
    const int ntap = ks * ks;
    const int nchunk = (CI + 7) / 8, ntile = (CO + KX_COT - 1) / KX_COT;
    const int64_t total = (int64_t)nchunk * ntile * ntap * KX_COT * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        const int ch = (int)(r % 8);
        r /= 8;
        const int co = (int)(r % KX_COT);
        r /= KX_COT;
        const int tap = (int)(r % ntap);
        r /= ntap;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int o = tile * KX_COT + co, i = chunk * 8 + ch;
        float v = 0.f;
        if (o < CO && i < CI) {
            if (!backward) v = w[((int64_t)o * cin + i) * ntap + tap];
            else v = w[((int64_t)i * cin + o) * ntap + (ntap - 1 - tap)];
        }
        v *= w_scale;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        const int64_t base = (((((int64_t)chunk * ntile + tile) * ntap + tap) * 2) * KX_COT + co) * 8 + ch;
        bank[base] = __builtin_bit_cast(unsigned short, h);
        bank[base + KX_COT * 8] = __builtin_bit_cast(unsigned short, l);
    }
}

template <int KS, bool ACC, bool OM>
__global__ void __launch_bounds__(256, 2) conv_kxk_x3_kernel(ConvArgs p, float w_inv_scale) {
    constexpr int PR = KX_TR + KS - 1, PC = KX_TC + KS - 1, NPOS = PR * PC, NTAP = KS * KS;
    constexpr int P_BYTES = 2 * NPOS * 16, W_BYTES = NTAP * 2 * KX_COT * 16;
    constexpr int NWQ = (W_BYTES + 4095) / 4096;  // 16-byte pieces of the filter slice per thread
    constexpr int NPQ = (NPOS + 255) / 256;       // patch positions per thread
    static_assert(P_BYTES + W_BYTES + 16 <= 65536, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_BYTES + W_BYTES + 16];
    unsigned char* Pl = smem;             // [part][pos][16 B]
    unsigned char* Wl = smem + P_BYTES;   // [tap][part][co][16 B]
    float* Ml = reinterpret_cast<float*>(smem + P_BYTES + W_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n = blockIdx.z / ksplit, split = blockIdx.z - n * ksplit;
    const int co0 = blockIdx.y * KX_COT;
    const int ntile = gridDim.y;
    const int in_plane = p.H * p.W;
    const int64_t out_plane = (int64_t)p.OH * p.OW;
    const float* __restrict__ xin = p.x + (int64_t)n * p.Cin * in_plane;
    const int x0 = (blockIdx.x % p.tiles_x) * KX_TC, y0 = (blockIdx.x / p.tiles_x) * KX_TR;

    unsigned p_byte[NPQ];
    bool pos_ok[NPQ];
#pragma unroll
    for (int q = 0; q < NPQ; ++q) {
        const int pos = tid + 256 * q;
        const int r = pos / PC, col = pos - r * PC;
        const int iy = y0 + r - p.pad, ix = x0 + col - p.pad;
        pos_ok[q] = pos < NPOS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        p_byte[q] = pos_ok[q] ? (unsigned)(iy * p.W + ix) * 4u : 0u;
    }
    float rp[NPQ][8];
    u32x4 rwq[NWQ];
    const unsigned char* __restrict__ bank = reinterpret_cast<const unsigned char*>(p.w6);
    auto load_chunk = [&](int ch) {
        asm volatile("" : "+s"(ch));
        const int c0 = ch * 8;
#pragma unroll
        for (int q = 0; q < NPQ; ++q) {
            if (q > 0 && tid + 256 * q >= NPOS) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int chn = min(c0 + c, p.Cin - 1);
                const char* plane = reinterpret_cast<const char*>(xin + (int64_t)chn * in_plane);
                const float v = *reinterpret_cast<const float*>(plane + p_byte[q]);
                rp[q][c] = (pos_ok[q] && c0 + c < p.Cin) ? v : 0.f;
            }
        }
        const unsigned char* src = bank + ((int64_t)ch * ntile + blockIdx.y) * W_BYTES;
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int off = tid * 16 + 4096 * q;
            if (off < W_BYTES) rwq[q] = *reinterpret_cast<const u32x4*>(src + off);
        }
    };
    auto publish_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < NPQ; ++q) {
            if (q > 0 && tid + 256 * q >= NPOS) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(rp[q][c]));
        }
        m = wave_max_nonneg(m);
        if (lane == 0) Ml[wave] = m;
    };
    // scale the staged chunk into [2^11, 2^12), split, write patch and filter slice; returns the factor that un-scales the sums
    auto store_chunk = [&]() {
        const float m = fmaxf(fmaxf(Ml[0], Ml[1]), fmaxf(Ml[2], Ml[3]));
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
        e = m > 0.f ? max(e, -100) : 11;
        const float sx = __builtin_bit_cast(float, (unsigned)(127 + 11 - e) << 23);
        const f32x2 sx2 = {sx, sx};
#pragma unroll
        for (int q = 0; q < NPQ; ++q) {
            const int pos = tid + 256 * q;
            if (pos >= NPOS) continue;
            u32x4 vh, vl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 v = f32x2{rp[q][2 * k], rp[q][2 * k + 1]} * sx2;
                const f16x2 h = __builtin_convertvector(v, f16x2);
                const f32x2 back = __builtin_convertvector(h, f32x2);
                vh[k] = __builtin_bit_cast(unsigned, h);
                vl[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, f16x2));
            }
            *reinterpret_cast<u32x4*>(Pl + pos * 16) = vh;
            *reinterpret_cast<u32x4*>(Pl + NPOS * 16 + pos * 16) = vl;
        }
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int off = tid * 16 + 4096 * q;
            if (off < W_BYTES) *reinterpret_cast<u32x4*>(Wl + off) = rwq[q];
        }
        return __builtin_bit_cast(float, (unsigned)(127 + e - 11) << 23) * w_inv_scale;
    };

    f32x16 acc[2], master[2];
    {
        const bool with_bias = p.bias != nullptr && p.ksplit <= 1;  // split: the finish kernel adds the bias
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                master[t][r] = with_bias ? p.bias[min(co, p.Cout - 1)] : 0.f;
            }
    }
    const unsigned char* a_base = Wl + j * 16;
    const unsigned char* b_base = Pl + (wave * PC + j) * 16;
    auto tap_bytes = [&](int tap, unsigned& ab, unsigned& bb) {
        const int ky = tap / KS, kx = tap - ky * KS;
        ab = (unsigned)tap * 2048u;
        bb = (unsigned)(ky * PC + kx) * 16u;
    };
    auto kstep = [&](int s) {  // taps 2 s (lane half 0) and 2 s + 1 (lane half 1); step 0 starts the chunk's sums from zero
        unsigned ab, bb;
        tap_bytes(2 * s + half, ab, bb);
        f16x8 b[2], a[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) b[part] = *reinterpret_cast<const f16x8*>(b_base + part * NPOS * 16 + bb);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 2; ++part) a[t][part] = *reinterpret_cast<const f16x8*>(a_base + ab + part * 1024 + t * 512);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[0], s == 0 ? zero : acc[t], 0, 0, 0);  // smallest terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };
    auto kstep_last = [&]() {  // the odd last tap alone: K = 8, lane half h takes channels 4 h .. 4 h + 3
        unsigned ab, bb;
        tap_bytes(NTAP - 1, ab, bb);
        f16x4 b[2], a[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) b[part] = *reinterpret_cast<const f16x4*>(b_base + part * NPOS * 16 + bb + half * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                a[t][part] = *reinterpret_cast<const f16x4*>(a_base + ab + part * 1024 + t * 512 + half * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };

    const int nchunks_all = (p.Cin + 7) / 8;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int ch_begin = split * cps;
    const int nchunks = min(nchunks_all, ch_begin + cps);
    if (ch_begin < nchunks) load_chunk(ch_begin);
    for (int ch = ch_begin; ch < nchunks; ++ch) {
        publish_max();
        __syncthreads();  // every wave is done with the previous chunk's LDS; this chunk's maxima are visible
        const float inv = store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk(ch + 1);  // in flight during this chunk's products
#pragma unroll
        for (int s = 0; s < NTAP / 2; ++s) kstep(s);
        kstep_last();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) master[t][r] = fmaf(acc[t][r], inv, master[t][r]);  // un-scale (power of two: exact) + fold
    }

    // epilogue: lane = pixel (y0 + wave, x0 + j); register r = output channel (r&3) + 8 (r>>2) + 4 half of block t
    const int oy = y0 + wave, ox = x0 + j;
    if (p.ksplit > 1) {  // un-scaled partial sums, added in split order by conv_splitk_finish_kernel
        if (oy < p.OH && ox < p.OW) {
            float* wsp = p.ws + ((int64_t)blockIdx.z * p.Cout + co0 + 4 * half) * out_plane + (int64_t)oy * p.OW + ox;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                    if (co0 + cr + 4 * half < p.Cout) wsp[(int64_t)cr * out_plane] = master[t][r];
                }
        }
        return;
    }
    if (oy < p.OH && ox < p.OW) {
        const int64_t lane_off = ((int64_t)n * p.Cout + co0 + 4 * half) * out_plane + (int64_t)oy * p.OW + ox;
        float* __restrict__ yl = p.y + lane_off;
        const float* __restrict__ oml = OM ? p.omask + lane_off : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float prev[16], msk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                const int64_t o = (co0 + cr + 4 * half < p.Cout) ? (int64_t)cr * out_plane : 0;
                prev[r] = 0.f;
                msk[r] = 1.f;
                if constexpr (ACC) prev[r] = yl[o];
                if constexpr (OM) msk[r] = oml[o];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = t * 32 + (r & 3) + 8 * (r >> 2);
                if (co0 + cr + 4 * half < p.Cout) {
                    float v = master[t][r] + prev[r];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    yl[(int64_t)cr * out_plane] = msk[r] > 0.f ? v : 0.f;
                }
            }
        }
    }
}

// Split of the input-channel loop when the output grid cannot fill the 512 workgroup slots (2 per CU): e.g. the backward pass
// of NIN's conv2 produces 96 channels = 2 channel tiles x 128 pixel tiles.
static int kxk_choose_split(const ConvArgs& a, int n) {
    const int64_t tiles = (int64_t)((a.OW + KX_TC - 1) / KX_TC) * ((a.OH + KX_TR - 1) / KX_TR);
    const int64_t wgs = tiles * ((a.Cout + KX_COT - 1) / KX_COT) * split_batch_hint();  // planned frames (see conv_x3w.hip)
    const int nchunks = (a.Cin + 7) / 8;
    int ks = (int)(512 / (wgs > 0 ? wgs : 1));
    if (ks > nchunks / 4) ks = nchunks / 4;
    if (ks > 8) ks = 8;
    return ks < 2 ? 1 : ks;
}

template <int KS>
static int conv_kxk_x3_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream) {
    ConvArgs p = a;
    p.tiles_x = (a.OW + KX_TC - 1) / KX_TC;
    const int64_t tiles = (int64_t)p.tiles_x * ((a.OH + KX_TR - 1) / KX_TR);
    const int ks = a.ws ? kxk_choose_split(a, n) : 1;
    p.ksplit = ks;
    dim3 grid((unsigned)tiles, (unsigned)((a.Cout + KX_COT - 1) / KX_COT), (unsigned)(n * ks));
    const bool acc = ks == 1 && a.accumulate != 0, om = ks == 1 && a.omask != nullptr;
    const float w_inv = 1.f / w_scale;
    if (acc && om) hipLaunchKernelGGL((conv_kxk_x3_kernel<KS, true, true>), grid, dim3(256), 0, stream, p, w_inv);
    else if (acc) hipLaunchKernelGGL((conv_kxk_x3_kernel<KS, true, false>), grid, dim3(256), 0, stream, p, w_inv);
    else if (om) hipLaunchKernelGGL((conv_kxk_x3_kernel<KS, false, true>), grid, dim3(256), 0, stream, p, w_inv);
    else hipLaunchKernelGGL((conv_kxk_x3_kernel<KS, false, false>), grid, dim3(256), 0, stream, p, w_inv);
    int rc = check_launch("conv_kxk_x3_kernel");
    if (rc || ks == 1) return rc;
    return conv_splitk_finish(a, n, ks, stream);
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_conv_kxk_x3_bank_bytes(int cout_produced, int cin_consumed, int ks) {
    if (cout_produced <= 0 || cin_consumed <= 0 || ks <= 0 || ks > 16 || cout_produced > (1 << 20) || cin_consumed > (1 << 20)) return 0;
    const size_t nchunk = (cin_consumed + 7) / 8, ntile = (cout_produced + KX_COT - 1) / KX_COT;
    return nchunk * ntile * (size_t)ks * ks * 2 * KX_COT * 16;
}

int maua_conv_pack_filters_kxk_x3(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, int ks, float w_scale,
                                  maua_stream_t stream) {
    MAUA_REQUIRE(w_oihw && (bank_fwd || bank_bwd) && cout > 0 && cin > 0 && cout <= (1 << 20) && cin <= (1 << 20) && ks > 0 && w_scale > 0.f, MAUA_E_INVAL,
                 "conv_pack_filters_kxk_x3: bad args");
    for (int backward = 0; backward < 2; ++backward) {
        void* bank = backward ? bank_bwd : bank_fwd;
        if (!bank) continue;
        hipLaunchKernelGGL(pack_kxk_x3_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (unsigned short*)bank, cout, cin, ks, backward, w_scale);
        int rc = check_launch("pack_kxk_x3_kernel");
        if (rc) return rc;
    }
    return MAUA_OK;
}

size_t maua_conv_kxk_x3_workspace_bytes(int n, int cin, int h, int w, int cout, int ks, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad) || ks <= 0 || ks > 16) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - ks + 1;
    a.OW = w + 2 * pad - ks + 1;
    if (a.OH <= 0 || a.OW <= 0) return 0;
    const int split = kxk_choose_split(a, n);
    return split > 1 ? (size_t)n * split * cout * a.OH * a.OW * sizeof(float) : 0;
}

int maua_conv_kxk_x3(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int ks, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && bank && y, MAUA_E_INVAL, "conv_kxk_x3: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && w_scale > 0.f, MAUA_E_INVAL, "conv_kxk_x3: bad dims");
    MAUA_REQUIRE(ks == 5, MAUA_E_UNSUPPORTED, "conv_kxk_x3: %dx%d filters (5x5 is built; 3x3 has conv_x3, 1x1 conv1x1_x3)", ks, ks);
    MAUA_REQUIRE(pad >= 0 && pad <= ks - 1, MAUA_E_UNSUPPORTED, "conv_kxk_x3: pad %d", pad);
    const int oh = h + 2 * pad - ks + 1, ow = w + 2 * pad - ks + 1;
    MAUA_REQUIRE(oh > 0 && ow > 0, MAUA_E_INVAL, "conv_kxk_x3: input smaller than the filter");
    MAUA_REQUIRE((int64_t)h * w < (1ll << 29), MAUA_E_UNSUPPORTED, "conv_kxk_x3: plane too large");
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
    a.OH = oh;
    a.OW = ow;
    a.pad = pad;
    a.relu = relu;
    a.accumulate = accumulate;
    a.ws = (workspace && workspace_bytes >= maua_conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, ks, pad)) ? (float*)workspace : nullptr;
    return conv_kxk_x3_launch<5>(a, n, w_scale, (hipStream_t)stream);
}

}  // extern "C"
