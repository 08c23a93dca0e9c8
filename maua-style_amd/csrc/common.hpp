// Shared helpers for libmaua_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/maua_hip.h"

namespace maua {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return MAUA_OK;
}

#define MAUA_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            maua::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

constexpr int kWave = 64;

// Kernels that ask for more than 64 KiB of dynamic LDS need hipFuncSetAttribute(MaxDynamicSharedMemorySize) once PER DEVICE (a process
// that moves to a second device must opt in again there); `done` = the call site's bit mask of devices served (relaxed atomics: a
// second thread at worst repeats the call).
inline hipError_t opt_in_dynamic_lds(const void* kernel, int bytes, unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    if (dev < 64 && ((__atomic_load_n(done, __ATOMIC_RELAXED) >> dev) & 1ull)) return hipSuccess;
    const hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc == hipSuccess && dev < 64) __atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELAXED);
    return rc;
}

// Extents every entry point accepts, checked BEFORE any shape arithmetic (found by the UBSan build: h + 2 * pad - 2 and
// friends overflow int for extents near INT_MAX): planes below 2^30 pixels (32-bit pixel offsets in the kernels), channel
// counts and batch sizes that keep every product the host forms inside int64.
inline bool conv_dims_ok(int64_t n, int64_t cin, int64_t h, int64_t w, int64_t cout, int64_t pad) {
    return n > 0 && n <= (1 << 16) && cin > 0 && cin <= (1 << 20) && cout > 0 && cout <= (1 << 20) && h > 0 && h <= (1 << 24) &&
           w > 0 && w <= (1 << 24) && pad >= 0 && pad <= 64 && h * w < (1ll << 30);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a workgroup of up to 1024 threads; result valid in thread 0.  `scratch` holds >= 16 doubles.
// Maximum of a NON-NEGATIVE float over the wave with DPP row shifts / row broadcasts (no LDS-pipe shuffles: lanes without
// a source receive 0, the identity for non-negative maxima); the result is wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_nonneg(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = dpp_max_nonneg<0x111, 0xf>(v);  // row_shr:1   (inclusive scan inside each row of 16 lanes)
    v = dpp_max_nonneg<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_nonneg<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_nonneg<0x118, 0xf>(v);  // row_shr:8
    v = dpp_max_nonneg<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v = dpp_max_nonneg<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3 ; lane 63 holds the maximum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += scratch[i];
    }
    __syncthreads();
    return r;
}

// Loss ledger: one record of LEDGER_STRIDE doubles per loss slot - {number of partial sums, scale, partial sums ...} - filled by
// the *_ledger entry points and summed (in finish_sum_kernel's order) by maua_loss_ledger_sum at the end of an evaluation.
constexpr int LEDGER_MAX = 4096;
constexpr int LEDGER_STRIDE = LEDGER_MAX + 2;

// Fixed-order second stage: one workgroup sums `n` per-block partials (doubles) in index order.
__global__ void finish_sum_kernel(const double* __restrict__ partial, int n, float scale, float* __restrict__ out);

// Arguments of the MFMA implicit-GEMM convolution (conv_mfma2.hip, conv_x6.hip); also used by gram.hip for gf += D F.
struct ConvArgs {
    const float* x;
    const float* mask;  // nullable: x is read as x * (mask > 0)
    const float* w;     // [KS*KS][Cin][Cout]
    const float* bias;  // nullable, [Cout]
    float* y;
    int Cin, H, W, Cout, OH, OW, pad;
    int tiles_x;
    int relu, accumulate;
    const void* w6;      // bf16x3 filter bank of conv_x6.hip (maua_conv_pack_filters_x6), else null
    const float* omask;  // nullable, output-shaped: result is zeroed where omask <= 0 (ReLU mask applied by the producer)
    int ksplit;          // conv_x6 split-K: > 1 -> blockIdx.z = n*ksplit + split, raw partial sums go to `ws`
    float* ws;           // [n][ksplit][Cout][OH][OW] partial sums (split-K only)
    int stagger;         // conv_x3w: start delay (units of 512 cycles) of the first-round workgroups in odd CU slots
    unsigned char* pool_codes;  // conv_x3w forward, nullable: the epilogue applies ReLU and the 2x2 / 2 max pool that follows,
                                // `y` is the POOLED map and these are its decision bytes (pool2x2_fwd_codes_kernel's)
    const void* dbank;   // conv_x3w, nullable: packed Cout x Cout matrix D (maua_conv_pack_dmat_x3w); the kernel adds D . omask
    const float* dinv;   //   (the Gram backward of the style loss on this layer's output) to its sums; dinv[0] = 1 / scale of the bank
    const unsigned char* in_codes;  // conv_x3w backward, nullable: `x` is the pooled map [Cin][H/2][W/2] of a 2x2 / 2 max pool and these are its
    int in_code_mask;               //   decision bytes; the kernel reads x as the pool's backward pass over them: element (y, x) =
                                    //   pooled[y/2][x/2] if (code & in_code_mask) == 2 (y & 1) + (x & 1) else 0  (mask 7: ReLU bit honoured, 3: not)
    int cot_inner;       // conv_x3w / conv_x3q: 1 = the channel tiles of a pixel tile are neighbours in dispatch order on ONE XCD (they run at the
                         //   same time and share the patch in that XCD's L2), 0 = blockIdx.y is the channel tile (a tile's channel tiles run gridDim.x
                         //   workgroups apart).  Which workgroup computes which (pixel tile, channel tile) changes no bit.
    unsigned* arrive;    // split-K finished INSIDE the launch (conv_x3q / conv_x3w, round 6), nullable: one arrival counter per (image, channel
                         //   tile, pixel tile), all zero between launches; the workgroup that draws the last ticket of its tile adds the other
                         //   splits' slabs to its own sums in split order and runs the one-pass epilogue (no conv_splitk_finish launch)
};
// conv_api.hip: the arrival counters of the workspace the calling host thread armed (maua_conv_arm_workspace) if `workspace` is that one,
// else null.  ARRIVE_COUNTERS words; a launch that needs more, or whose workspace is another one, finishes its split in a second launch.
constexpr int ARRIVE_COUNTERS = 4096;
unsigned* armed_counters(const void* workspace);
// conv_api.hip: the library's tuning constants (plan.py lists them; maua_set_tuning sets them, nothing reads the environment)
double tuning(const char* name, double dflt);
int split_batch_hint();  // conv_api.hip: frames per launch the caller plans with (split-K cost models), >= 1
int conv_mfma_dispatch(const ConvArgs& a, int ks, int n, hipStream_t stream);
int conv3x3_few_out(const ConvArgs& a, int n, hipStream_t stream);
int conv_splitk_finish(const ConvArgs& a, int n, int ksplit, hipStream_t stream);
int conv_splitk_finish_pool(const ConvArgs& a, int n, int ksplit, hipStream_t stream);  // ... + ReLU + 2x2 max pool: a.y = pooled map, a.pool_codes
int conv_x3_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream);  // conv_x3.hip  // y = act(bias + sum of a.ws partials) ...
int conv_x3w_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream);  // conv_x3w.hip (16-channel chunks, a.Cin % 16 == 0)
bool conv_x3w_supports(const ConvArgs& a);
int conv_x3q_launch(const ConvArgs& a, int n, float w_scale, hipStream_t stream);  // conv_x3q.hip (32-channel chunks on 16x16x32, a.Cin % 32 == 0)
bool conv_x3q_supports(const ConvArgs& a);
int conv1x1_x3_launch(const ConvArgs& a, const float* xshift, int n, hipStream_t stream);  // conv1x1_x3.hip (a.w = [Cout][Cin], a.H*a.W pixels)
size_t conv1x1_x3_workspace(int n, int cin, int64_t hw, int cout);
int conv_mfma2_choose_split(const ConvArgs& a, int ks, int n);                     // 1 = no split  // 3x3, stride 1, Cout <= 4 (conv_direct.hip)
int conv_direct_fwd(const float* x, const float* mask, const float* wf, const float* bias, float* y, int n, int cin, int h,
                    int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int relu, int accumulate,
                    hipStream_t stream);
int conv_direct_bwd(const float* gy, const float* mask, const float* w_oihw, const float* omask, float* gx, int n, int cin,
                    int h, int w, int cout, int oh, int ow, int kh, int kw, int stride, int pad, int accumulate,
                    hipStream_t stream);

inline int reduce_blocks(int64_t count, int per_block) {
    int64_t b = (count + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace maua
