// C-ABI entry points of the convolution family: route to the MFMA implicit-GEMM path (stride 1, k in {1,3,5}) or
// to the direct path (everything else).
#include <cstdlib>
#include <cstring>

#include "common.hpp"

using namespace maua;

namespace maua {
// Frames the caller plans to evaluate per launch (see maua_set_split_batch_hint): the split-K cost models count
// `frames` images' worth of workgroups, whatever the batch size of the launch at hand.  Per host thread, like the error
// message: two jobs with different plans in one process (one thread each) do not see each other's value.
static thread_local int g_split_batch_hint = 1;
int split_batch_hint() { return g_split_batch_hint; }
// The workspace whose split-K launches may finish inside the launch, and its arrival counters (maua_conv_arm_workspace).  Per host thread:
// a workspace belongs to one stream of launches at a time anyway (its slabs), and so do its counters.
static thread_local const void* g_armed_ws = nullptr;
static thread_local unsigned* g_armed_counters = nullptr;
unsigned* armed_counters(const void* workspace) { return workspace && workspace == g_armed_ws ? g_armed_counters : nullptr; }
}  // namespace maua

namespace maua {
// Tuning constants of the library: routing thresholds and experiment switches that used to be `getenv` calls spread over csrc/.  The host
// side (plan.py) owns the list and sets the values through maua_set_tuning when it loads the library; a name that was never set reads
// as the caller's default.  Process-wide, read at every use (a few string compares per launch), not thread-safe against concurrent sets.
static const char* const g_tuning_names[] = {"conv_few_out", "few_out_ks4_below", "x3w_ks", "x3w_stagger", "x3q_ks", "x3q_min_fill", "x3q_min_chunks",
                                             "x3p_ks", "x3p_groups", "x3p_min_fill", "x3p_min_items", "x6_persist", "gram_x3", "gram_bwd_x3",
                                             "gram_t128", "gram_t128_min_hw", "p1_order", "lbfgs_vec", "lbfgs_tri", "finish_in_launch_max_ks", "x3p_order", "cot_inner"};
constexpr int kTunings = sizeof(g_tuning_names) / sizeof(g_tuning_names[0]);
static double g_tuning_values[kTunings];
static bool g_tuning_set[kTunings];
static int tuning_index(const char* name) {
    for (int i = 0; i < kTunings; ++i)
        if (name && strcmp(name, g_tuning_names[i]) == 0) return i;
    return -1;
}
double tuning(const char* name, double dflt) {
    const int i = tuning_index(name);
    return i >= 0 && g_tuning_set[i] ? g_tuning_values[i] : dflt;
}
}  // namespace maua

static inline bool mfma_geometry(int kh, int kw, int stride) {
    return stride == 1 && kh == kw && (kh == 1 || kh == 3 || kh == 5);
}
// a 3x3 pass that produces <= 4 channels from many (conv1_1 backward-data) is vector-ALU work, not an MFMA tile
static int conv_route(const ConvArgs& a, int ks, int n, hipStream_t stream) {
    const bool few_out = tuning("conv_few_out", 1) != 0;  // (experiment switch)
    if (ks == 3 && a.Cout <= 4 && a.Cin >= 16 && few_out) return conv3x3_few_out(a, n, stream);
    return conv_mfma_dispatch(a, ks, n, stream);
}

extern "C" {

int maua_set_tuning(const char* name, double value) {
    const int i = maua::tuning_index(name);
    MAUA_REQUIRE(i >= 0, MAUA_E_INVAL, "set_tuning: unknown constant %s", name ? name : "(null)");
    maua::g_tuning_values[i] = value;
    maua::g_tuning_set[i] = true;
    return MAUA_OK;
}
double maua_get_tuning(const char* name, double dflt) { return maua::tuning(name, dflt); }

int maua_conv_arm_workspace(void* workspace, void* counters, size_t counter_bytes, int zero, maua_stream_t stream) {
    if (!workspace || !counters) {  // disarm
        maua::g_armed_ws = nullptr;
        maua::g_armed_counters = nullptr;
        return MAUA_OK;
    }
    MAUA_REQUIRE(counter_bytes >= (size_t)maua::ARRIVE_COUNTERS * 4 && ((size_t)counters & 15) == 0, MAUA_E_INVAL,
                 "conv_arm_workspace: needs %d bytes of 16-byte aligned counters", maua::ARRIVE_COUNTERS * 4);
    const hipError_t rc = zero ? hipMemsetAsync(counters, 0, (size_t)maua::ARRIVE_COUNTERS * 4, (hipStream_t)stream) : hipSuccess;
    if (rc != hipSuccess) {
        set_error("conv_arm_workspace: hipMemsetAsync: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    maua::g_armed_ws = workspace;
    maua::g_armed_counters = (unsigned*)counters;
    return MAUA_OK;
}

void maua_set_split_batch_hint(int frames) { maua::g_split_batch_hint = frames > 0 ? frames : 1; }
int maua_get_split_batch_hint(void) { return maua::g_split_batch_hint; }

size_t maua_conv_workspace_bytes(int n, int cin, int h, int w, int cout, int kh, int kw, int stride, int pad) {
    if (!conv_dims_ok(n, cin, h, w, cout, pad) || kh <= 0 || kw <= 0 || kh > 64 || kw > 64 || stride <= 0 || stride > 64) return 0;
    if (!mfma_geometry(kh, kw, stride) || h + 2 * pad < kh || w + 2 * pad < kw) return 0;
    ConvArgs a{};
    a.Cin = cin;
    a.Cout = cout;
    a.OH = h + 2 * pad - kh + 1;
    a.OW = w + 2 * pad - kw + 1;
    if (kh == 3 && cout <= 4 && cin >= 16) return 0;  // few-output-channel kernel: no split
    const int ks = conv_mfma2_choose_split(a, kh, n);
    return ks > 1 ? (size_t)n * ks * cout * a.OH * a.OW * sizeof(float) : 0;
}

int maua_conv2d_fwd(const float* x, const float* in_mask, const float* wf, const float* bias, float* y, int n, int cin,
                    int h, int w, int cout, int kh, int kw, int stride, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(x && wf && y, MAUA_E_INVAL, "conv2d_fwd: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && kh > 0 && kw > 0 && kh <= 64 && kw <= 64 && stride > 0 && stride <= 64, MAUA_E_INVAL,
                 "conv2d_fwd: bad dims n=%d cin=%d cout=%d h=%d w=%d k=%dx%d s=%d p=%d", n, cin, cout, h, w, kh, kw, stride,
                 pad);
    const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
    MAUA_REQUIRE(h + 2 * pad >= kh && w + 2 * pad >= kw, MAUA_E_UNSUPPORTED, "conv2d_fwd: input %dx%d smaller than filter", h,
                 w);
    MAUA_REQUIRE((int64_t)h * w < (1ll << 31) && (int64_t)oh * ow < (1ll << 31), MAUA_E_UNSUPPORTED,
                 "conv2d_fwd: plane too large");
    if (mfma_geometry(kh, kw, stride)) {
        ConvArgs a{};
        a.x = x;
        a.mask = in_mask;
        a.w = wf;
        a.bias = bias;
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
        a.ws = (workspace && workspace_bytes >= maua_conv_workspace_bytes(n, cin, h, w, cout, kh, kw, stride, pad))
                   ? (float*)workspace : nullptr;
        return conv_route(a, kh, n, (hipStream_t)stream);
    }
    return conv_direct_fwd(x, in_mask, wf, bias, y, n, cin, h, w, cout, oh, ow, kh, kw, stride, pad, relu, accumulate,
                           (hipStream_t)stream);
}

int maua_conv2d_bwd_data(const float* gy, const float* out_mask, const float* wb, const float* w_oihw,
                         const float* in_relu_mask, float* gx, int n, int cin, int h, int w, int cout, int kh, int kw,
                         int stride, int pad, int accumulate, void* workspace, size_t workspace_bytes, maua_stream_t stream) {
    MAUA_REQUIRE(gy && gx && (wb || w_oihw), MAUA_E_INVAL, "conv2d_bwd_data: null pointer");
    MAUA_REQUIRE(conv_dims_ok(n, cin, h, w, cout, pad) && kh > 0 && kw > 0 && kh <= 64 && kw <= 64 && stride > 0 && stride <= 64, MAUA_E_INVAL,
                 "conv2d_bwd_data: bad dims");
    const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
    MAUA_REQUIRE(h + 2 * pad >= kh && w + 2 * pad >= kw, MAUA_E_UNSUPPORTED, "conv2d_bwd_data: input smaller than filter");
    if (mfma_geometry(kh, kw, stride) && wb) {
        // gx = conv(gy masked, flipped/transposed bank) with padding k-1-p: a forward conv over the gradient
        ConvArgs a{};
        a.x = gy;
        a.mask = out_mask;
        a.w = wb;
        a.bias = nullptr;
        a.omask = in_relu_mask;
        a.y = gx;
        a.Cin = cout;
        a.H = oh;
        a.W = ow;
        a.Cout = cin;
        a.OH = h;
        a.OW = w;
        a.pad = kh - 1 - pad;
        a.relu = 0;
        a.accumulate = accumulate;
        MAUA_REQUIRE(a.pad >= 0, MAUA_E_UNSUPPORTED, "conv2d_bwd_data: pad %d > k-1", pad);
        // backward-data = forward geometry with the channel roles exchanged on the gradient's plane
        a.ws = (workspace && workspace_bytes >= maua_conv_workspace_bytes(n, cout, oh, ow, cin, kh, kw, 1, a.pad))
                   ? (float*)workspace : nullptr;
        return conv_route(a, kh, n, (hipStream_t)stream);
    }
    MAUA_REQUIRE(w_oihw, MAUA_E_INVAL, "conv2d_bwd_data: the direct path needs the OIHW weights");
    return conv_direct_bwd(gy, out_mask, w_oihw, in_relu_mask, gx, n, cin, h, w, cout, oh, ow, kh, kw, stride, pad, accumulate,
                           (hipStream_t)stream);
}

}  // extern "C"
