/*
 * maua_hip.h - C ABI of libmaua_hip.so: the MI355X (gfx950) kernels behind the maua-style
 * image-optimisation hot path.
 *
 * The reference (JCBrouwer/maua-style) has no FFI layer: its boundary is the Python surface of
 * optim.py / loss.py / models.py and every FLOP is a PyTorch op.  Each entry point below replaces
 * the torch op(s) the reference issues at the cited lines (paths relative to the reference root).
 * The host side (the Python modules in maua-style_amd/) binds these symbols with ctypes; see INTEGRATION.md for the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a caller-allocated DEVICE pointer (e.g. torch.Tensor.data_ptr() of a
 *    contiguous fp32 ROCm tensor) unless a parameter says "host"; the library never allocates,
 *    frees or retains caller memory;
 *  - tensors are float32, NCHW, contiguous;
 *  - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is enqueued
 *    asynchronously on it, nothing synchronises, so calls can be captured into a hipGraph;
 *  - return value: 0 = ok, <0 = invalid argument / unsupported shape (MAUA_E_*), >0 = hipError_t
 *    of the failed launch; maua_last_error() gives a thread-local message;
 *  - per host thread: the error message, the split-K batch hint (maua_set_split_batch_hint) and the armed workspace
 *    (maua_conv_arm_workspace), so concurrent jobs on their own threads and streams do not interfere through them.  PROCESS-WIDE mutable state, exactly two setters: maua_set_tuning (the planner's
 *    constants, set once when the host side loads the library) and maua_conv_x3p_set_max_groups (tests); neither changes a result bit,
 *    both are read at launch time - set them before the first launch, not while another thread launches or captures;
 *  - reductions are fixed-order (no float atomics): reruns are bit-identical, like the reference
 *    at a fixed thread count.
 */
#ifndef MAUA_HIP_H
#define MAUA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* maua_stream_t; /* hipStream_t */

#define MAUA_OK 0
#define MAUA_E_INVAL (-1)       /* null pointer, non-positive size, bad flag */
#define MAUA_E_UNSUPPORTED (-2) /* shape outside what the kernels implement */
#define MAUA_E_WORKSPACE (-3)   /* workspace too small */

int maua_abi_version(void);
const char* maua_last_error(void);

/* ---- convolution: nn.Conv2d (+ nn.ReLU(inplace)) forward, models.py:129-130 (VGG), :83-110 (NIN);
 *      backward-data = autograd of the same (optim.py:213), weights frozen (models.py:444) ---------- */

/* Re-layout OIHW weights (state-dict order) into the two filter banks the kernels read:
 *   wf[ky*KW+kx][ci][co] = w[co][ci][ky][kx]                      (forward)
 *   wb[(KH-1-ky)*KW+(KW-1-kx)][co][ci] = w[co][ci][ky][kx]        (backward-data, stride 1)
 * Each bank has cout*cin*kh*kw floats.  Either destination may be NULL. */
int maua_conv_pack_filters(const float* w_oihw, float* wf, float* wb, int cout, int cin, int kh, int kw,
                           maua_stream_t stream);

/* y[n][co][oy][ox] = act( bias[co] + sum_{ci,ky,kx} x[n][ci][oy*s-p+ky][ox*s-p+kx] * w[co][ci][ky][kx] ) (+ y if accumulate)
 *  wf: forward bank from maua_conv_pack_filters.  in_mask (nullable, same shape as x): x is read as
 *  x * (in_mask > 0) - the fused threshold_backward of the ReLU in front of a backward-data pass.
 *  bias nullable.  relu: 0/1.  accumulate: 0/1 (add into y instead of overwriting).
 *  stride 1 with k in {1,3,5} runs on the fp32 MFMA path; other geometry on the direct path.
 *  workspace (nullable): maua_conv_workspace_bytes(...) bytes let layers whose output grid cannot fill the chip (deep
 *  1x1 layers on small maps) split the channel loop over workgroups; the partial sums are added in a fixed order. */
size_t maua_conv_workspace_bytes(int n, int cin, int h, int w, int cout, int kh, int kw, int stride, int pad);
int maua_conv2d_fwd(const float* x, const float* in_mask, const float* wf, const float* bias, float* y, int n, int cin,
                    int h, int w, int cout, int kh, int kw, int stride, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream);

/* gx = conv_transpose(gy * (out_mask > 0), w): gradient w.r.t. the conv input.  gy/out_mask: [n][cout][oh][ow] (mask
 * nullable, it is the saved ReLU output of this conv), gx: [n][cin][h][w].  wb: backward bank (stride 1) or the OIHW
 * weights themselves (stride > 1, direct path: pass w_oihw and wb = NULL).  in_relu_mask (nullable, shape of gx): the
 * result is zeroed where it is <= 0 - the threshold_backward of the ReLU that produced this conv's INPUT, applied by the
 * producer of the gradient so that the next backward pass needs no mask while staging.
 * workspace (nullable): maua_conv_workspace_bytes(n, cout, oh, ow, cin, kh, kw, 1, k-1-pad) bytes (the same pass seen as
 * a forward convolution over the gradient). */
int maua_conv2d_bwd_data(const float* gy, const float* out_mask, const float* wb, const float* w_oihw,
                         const float* in_relu_mask, float* gx, int n, int cin, int h, int w, int cout, int kh, int kw,
                         int stride, int pad, int accumulate, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- 3x3 stride-1 convolution at fp32 accuracy on the bf16 matrix cores (three-way bf16 split of both operands, six
 *      MFMAs per product block; conv_x6.hip).  Same math as maua_conv2d_fwd / maua_conv2d_bwd_data for k = 3, s = 1. ---- */
/* Bytes of one pre-split bank for a pass that PRODUCES `cout_produced` channels from `cin_consumed` channels. */
size_t maua_conv_x6_bank_bytes(int cout_produced, int cin_consumed);
/* OIHW fp32 weights [cout][cin][3][3] -> forward bank (maua_conv_x6_bank_bytes(cout, cin)) and backward-data bank
 * (maua_conv_x6_bank_bytes(cin, cout), taps flipped, roles swapped).  Either destination may be NULL. */
int maua_conv_pack_filters_x6(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, maua_stream_t stream);
/* y[n][cout][h+2p-2][w+2p-2] = act(bias + conv3x3(x, bank)) (+ y if accumulate), then zeroed where out_relu_mask <= 0
 * (nullable; the ReLU mask a backward-data pass applies on behalf of its consumer).  Forward: bank_fwd, pad p.
 * Backward-data of a pad-p conv: x = gradient w.r.t. the conv output, bank_bwd, cin/cout exchanged, pad 2-p.
 * workspace (nullable): maua_conv_x6_workspace_bytes(...) bytes let small output grids split the channel loop over
 * several workgroups (partial sums added in a fixed order by a second kernel: still deterministic). */
size_t maua_conv_x6_workspace_bytes(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_x6(const float* x, const void* bank, const float* bias, const float* out_relu_mask, float* y, int n,
                    int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream);

/* The image layer in the same exact bf16x6 arithmetic without the general kernel's padding: a 3x3 stride-1 convolution that consumes
 * 1-3 channels (`nn.Conv2d(3, 64, 3, padding=1)` + `nn.ReLU`, models.py:129-130: conv1_1) with K = the 9 cin (channel, tap) pairs in two
 * K = 16 matrix steps, filters in registers, pixels gathered from the image - bound by writing the activation (conv_img.hip).
 * Bank: maua_conv_image_bank_bytes(cout, cin) bytes, packed from the OIHW weights AND the bias (nullable) once per weight set: the bias
 * rides in the first unused pair of the padded K against a pixel value of 1.  Forward only.  maua_conv_image_supported: the geometry check -
 * the kernel's offsets are 32-bit, its store descriptor spans 64 output planes: planes of fewer than 2^24 pixels (a 4096 x 4096 image is one
 * pixel past that; callers route such layers to maua_conv3x3_x6, the same products).  maua_conv_image_gram_slabs returns 0 for them. */
int maua_conv_image_supported(int n, int cin, int h, int w, int cout, int pad);
size_t maua_conv_image_bank_bytes(int cout, int cin);
int maua_conv_pack_filters_image(const float* w_oihw, const float* bias, void* bank, int cout, int cin, maua_stream_t stream);
int maua_conv3x3_image(const float* x, const void* bank, float* y, int n, int cin, int h, int w, int cout, int pad, int relu,
                       maua_stream_t stream);
/* The same launch (one image, 64 output channels, ReLU) which also leaves the Gram matrix of the activation it writes
 * (`torch.mm(x, x.T)` of loss.GramMatrix.forward, loss.py:91, on relu1_1) as split-K slabs: maua_conv_image_gram_slabs(h, w, pad) slabs of
 * 64 x 64 floats in `gram_slabs` (the layout of maua_gram_partial's workspace for C = 64: upper channel blocks filled), one per
 * workgroup, to be folded and finished by maua_gram_partial_batch / maua_gram_finish_mse_batch with that slab count - the 268 MB of
 * relu1_1 at 1024 x 1024 are not read back for it.  The activation is bit-identical to maua_conv3x3_image's. */
int maua_conv_image_gram_slabs(int h, int w, int pad);
int maua_conv3x3_image_gram(const float* x, const void* bank, float* y, float* gram_slabs, int cin, int h, int w, int pad, maua_stream_t stream);

/* ---- the same convolution on the fp16 matrix cores: a power-of-two-scaled fp32 value as two fp16 parts (22 of 24
 *      significant bits, representation error 7e-8 of the result: below fp32 accumulation noise), three MFMAs per
 *      product block (conv_x3.hip).  Filters are scaled once per layer by `w_scale` (a power of two with
 *      |w| * w_scale < 64, chosen by the caller when packing); activations are scaled inside the kernel per workgroup
 *      and 8-channel chunk.  Arguments as maua_conv3x3_x6. ---- */
size_t maua_conv_x3_bank_bytes(int cout_produced, int cin_consumed);
int maua_conv_pack_filters_x3(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                              maua_stream_t stream);
size_t maua_conv_x3_workspace_bytes(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_x3(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                    int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                    size_t workspace_bytes, maua_stream_t stream);

/* The same fp16x3 arithmetic on the second kernel structure (conv_x3w.hip): K chunks of 16 input channels (every tap a full
 * K = 16 MFMA step), four 32x32 accumulators per wave, two workgroups per CU.  Replaces the same torch calls as
 * maua_conv3x3_x3 (nn.Conv2d 3x3 + ReLU forward, models.py:129-130, and its backward-data) for layers with cin % 16 == 0
 * and planes of at most 2^25 pixels (maua_conv_x3w_supported); bank layout [chunk16][cout tile][tap][part][octet][co][8 ch]. */
size_t maua_conv_x3w_bank_bytes(int cout_produced, int cin_consumed);
int maua_conv_pack_filters_x3w(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                               maua_stream_t stream);
int maua_conv_x3w_supported(int cin, int h, int w, int pad);
size_t maua_conv_x3w_workspace_bytes(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_x3w(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream);

/* conv3x3 + bias + ReLU + the 2x2 stride-2 max pool that follows it (`nn.Conv2d` / `nn.ReLU` / `nn.MaxPool2d(2, 2)`,
 * models.py:120-130) in one launch: the full-size activation is not written at all - `pooled` (n, cout, oh / 2, ow / 2) and
 * the decision bytes of maua_pool2x2_fwd_codes come out of the convolution's epilogue; bit-identical to the two separate
 * launches.  workspace (nullable) as for maua_conv3x3_x3w: where maua_conv_x3w_split(...) - the number of channel-loop splits
 * maua_conv3x3_x3w would use for this geometry under the current batch hint - is above 1 and the workspace holds the slabs, the
 * ReLU and the pool happen in the pass that adds the slabs (two launches; still no full-size activation, still the same bits). */
int maua_conv_x3w_split(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_x3w_relu_pool(const float* x, const void* bank, float w_scale, const float* bias, float* pooled,
                               unsigned char* codes, int n, int cin, int h, int w, int cout, int pad, void* workspace,
                               size_t workspace_bytes, maua_stream_t stream);

/* The backward-data pass of a 3x3 layer fused with the Gram backward of the style loss that sits on the layer's INPUT
 * activation F (reference: autograd of `torch.mm(x, y.T)` in GramMatrix.forward, loss.py:91, summed by autograd with the
 * convolution's input gradient): y = [F > 0] * (conv3x3(x; backward bank) + D . F), D = the symmetric cout x cout matrix
 * grad_scale (G - T) of maua_mse_fwd_bwd / maua_gram_fwd_mse_ledger, packed by maua_conv_pack_dmat_x3w into `dmat_bank`
 * (maua_conv_x3w_dmat_bank_bytes(cout) bytes; dmat_inv_scale = one device float the packer writes; with n > 1 images every
 * image has its own matrix: n consecutive banks, n consecutive scales).  Saves the separate
 * maua_gram_bwd pass (read F, read-modify-write y).  Needs cout % 16 == 0, pad = 1 geometry (output plane = input plane),
 * no covariance centring; `feature_map` is both the ReLU mask and the operand of D . F. */
size_t maua_conv_x3w_dmat_bank_bytes(int c);
int maua_conv_pack_dmat_x3w(const float* dmat, int c, void* bank, float* inv_scale_out, maua_stream_t stream);
/* ... of up to four layers in one launch (host arrays of `count` entries; the same banks and scales as the per-layer call) */
int maua_conv_pack_dmat_x3w_batch(int count, const float* const* dmats, const int* cs, void* const* banks, float* const* inv_scales_out,
                                  maua_stream_t stream);
int maua_conv3x3_x3w_gram(const float* x, const void* bank, float w_scale, const float* feature_map, const void* dmat_bank,
                          const float* dmat_inv_scale, float* y, int n, int cin, int h, int w, int cout, int pad, int accumulate,
                          void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* The backward-data pass of a 3x3 layer whose output went through ReLU + `nn.MaxPool2d(2, 2)` (models.py:120, 129-130), fused with
 * that pool's backward pass (autograd's max_pool2d_with_indices_backward + threshold_backward): `pooled_x` is the gradient w.r.t. the
 * POOLED map (n, cin, h / 2, w / 2), `codes` the pool's decision bytes (maua_conv3x3_x3w_relu_pool / maua_pool2x2_fwd_codes); the
 * kernel stages element (y, x) of the full-size gradient as pooled_x[y / 2][x / 2] where the byte names that corner (and, with
 * `honour_relu_bit`, the winning value was positive), 0 elsewhere - exactly what maua_pool2x2_bwd_codes would have written, which is
 * then never written nor read.  h, w = the full-size plane (even); bank = the layer's backward bank, cin = channels of the gradient.
 * out_relu_mask / dmat_bank / dmat_inv_scale (nullable) as in maua_conv3x3_x3w / maua_conv3x3_x3w_gram (with a bank, out_relu_mask is
 * the feature map F).  Bit-identical to maua_pool2x2_bwd_codes followed by maua_conv3x3_x3w / _gram. */
int maua_conv3x3_x3w_unpool(const float* pooled_x, const unsigned char* codes, int honour_relu_bit, const void* bank, float w_scale,
                            const float* out_relu_mask, const void* dmat_bank, const float* dmat_inv_scale, float* y, int n, int cin,
                            int h, int w, int cout, int pad, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* The same fp16x3 arithmetic on the third kernel structure (conv_x3q.hip, round 4): K chunks of 32 input channels on
 * v_mfma_f32_16x16x32_f16 (one k-step = one tap x 32 channels), one workgroup of EIGHT waves (512 threads, two per SIMD) per CU with all nine taps of a
 * chunk resident in LDS.  Same layer arithmetic as maua_conv3x3_x3w - `nn.Conv2d(cin, c, 3)` + `nn.ReLU(inplace=True)`,
 * models.py:129-130, and its backward-data pass - same arguments and flags, same split-K workspace protocol; needs
 * cin % 32 == 0 and planes of at most 2^24 pixels (maua_conv_x3q_supported); bank layout
 * [chunk32][cout tile][tap][part][octet 4][co][8 ch], same power-of-two filter scale as the other fp16x3 banks. */
size_t maua_conv_x3q_bank_bytes(int cout_produced, int cin_consumed);
int maua_conv_pack_filters_x3q(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, float w_scale,
                               maua_stream_t stream);
int maua_conv_x3q_supported(int cin, int h, int w, int pad);
size_t maua_conv_x3q_workspace_bytes(int n, int cin, int h, int w, int cout, int pad);
int maua_conv_x3q_split(int n, int cin, int h, int w, int cout, int pad);
/* 1 where this geometry fills the kernel's 256 workgroup slots (one per CU) well enough to beat conv_x3w's finer tiles - the host side's
 * routing rule, under the current batch hint; 0 otherwise. */
int maua_conv_x3q_preferred(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_x3q(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream);
/* ... with ReLU and the 2x2 / 2 max pool behind the layer in the epilogue (`nn.MaxPool2d(2, 2)`, models.py:120): arguments, outputs
 * (pooled map + decision bytes in the octet-interleaved layout) and split-K behaviour of maua_conv3x3_x3w_relu_pool. */
int maua_conv3x3_x3q_relu_pool(const float* x, const void* bank, float w_scale, const float* bias, float* pooled, unsigned char* codes,
                               int n, int cin, int h, int w, int cout, int pad, void* workspace, size_t workspace_bytes,
                               maua_stream_t stream);
/* ... as the backward-data pass that stages its input straight from the POOLED map's gradient and the pool's decision bytes
 * (maua_conv3x3_x3w_unpool without the Gram term: the layer the host routes here - conv4_4, 512 channels - carries no style loss on its input).
 * Bit-identical to maua_pool2x2_bwd_codes followed by maua_conv3x3_x3q. */
int maua_conv3x3_x3q_unpool(const float* pooled_x, const unsigned char* codes, int honour_relu_bit, const void* bank, float w_scale,
                            const float* out_relu_mask, float* y, int n, int cin, int h, int w, int cout, int pad, void* workspace,
                            size_t workspace_bytes, maua_stream_t stream);

/* The fourth structure (conv_x3p.hip, round 5): conv_x3q's workgroup made persistent.  One workgroup of eight waves per CU walks a
 * static share of the launch's work items - (image, 64-channel output tile, 16 x 32 pixel tile, K split) - as one stream of 32-channel
 * chunks: the first chunk of the next item is staged during the last chunk of the current one, and an item's epilogue (ReLU, mask, pool,
 * stores) rides among the MFMAs of the next item's first chunk.  Same layer arithmetic - `nn.Conv2d(cin, c, 3)` + `nn.ReLU(inplace=True)`
 * (+ `nn.MaxPool2d(2, 2)`), models.py:120, 129-130, and the backward-data pass autograd derives - same banks (maua_conv_pack_filters_x3q)
 * and, in one pass over the channels, the same bits as maua_conv3x3_x3q / _relu_pool / _unpool.  ONE entry point, every fused form an
 * argument (all nullable):
 *   in_codes + honour_relu_bit  x is the gradient of the POOLED map and these the pool's decision bytes (maua_conv3x3_x3w_unpool);
 *   bias, relu, out_relu_mask   as maua_conv3x3_x3q (no accumulation);
 *   dmat_bank + dmat_inv_scale  the Gram backward D . F of the style loss on out_relu_mask = F goes along (maua_conv3x3_x3w_gram; the
 *                               bank is maua_conv_pack_dmat_x3w's);
 *   pool_codes                  y is the POOLED map and these its decision bytes (maua_conv3x3_x3w_relu_pool; needs relu, no mask).
 * Needs cin % 32 == 0, cout % 64 == 0, cout <= 512 (the bias table in LDS), planes of at most 2^24 pixels and one image's output below 2 GiB
 * (maua_conv_x3p_supported).  workspace as for maua_conv3x3_x3q (maua_conv_x3p_workspace_bytes; maua_conv_x3p_split = the K splits a
 * launch would use, 1 = one pass). */
int maua_conv_x3p_supported(int cin, int h, int w, int cout, int pad);
int maua_conv_x3p_split(int n, int cin, int h, int w, int cout, int pad);
size_t maua_conv_x3p_workspace_bytes(int n, int cin, int h, int w, int cout, int pad);
/* 1 where the launch's work items deal out evenly enough over the 256 persistent workgroups (and its 16 x 32 tiles cover the plane
 * tightly enough) to beat conv_x3w's finer tiles - the host side's routing rule, under the current batch hint; 0 otherwise. */
int maua_conv_x3p_preferred(int n, int cin, int h, int w, int cout, int pad);
/* Tests: the workgroups a launch may use (a multiple of 8 from 8 on; anything else restores one per CU = 256), so that small shapes walk
 * several items per workgroup.  Returns the previous OVERRIDE (0 = none: the tuning constant x3p_groups decides), so that passing the
 * returned value back restores the state exactly.  Process-wide; results do not depend on it (a tile's arithmetic is the same
 * wherever it sits in a workgroup's list). */
int maua_conv_x3p_set_max_groups(int groups);
int maua_conv3x3_x3p(const float* x, const unsigned char* in_codes, int honour_relu_bit, const void* bank, float w_scale, const float* bias,
                     const float* out_relu_mask, const void* dmat_bank, const float* dmat_inv_scale, float* y, unsigned char* pool_codes,
                     int n, int cin, int h, int w, int cout, int pad, int relu, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- KS x KS stride-1 convolution in the same fp16x3 arithmetic (conv_kxk_x3.hip; KS = 5: NIN's conv2, models.py:86).
 *      Banks as for maua_conv_pack_filters_x3 with KS*KS taps; backward-data of a pad-p conv: bank_bwd, cin/cout exchanged,
 *      pad KS-1-p.  workspace (nullable) as for maua_conv3x3_x6: lets small output grids split the channel loop. ---- */
size_t maua_conv_kxk_x3_bank_bytes(int cout_produced, int cin_consumed, int ks);
int maua_conv_pack_filters_kxk_x3(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, int ks, float w_scale,
                                  maua_stream_t stream);
size_t maua_conv_kxk_x3_workspace_bytes(int n, int cin, int h, int w, int cout, int ks, int pad);
int maua_conv_kxk_x3(const float* x, const void* bank, float w_scale, const float* bias, const float* out_relu_mask, float* y,
                     int n, int cin, int h, int w, int cout, int ks, int pad, int relu, int accumulate, void* workspace,
                     size_t workspace_bytes, maua_stream_t stream);

/* ---- 1x1 convolution / channel-mixing product y[co][p] (+)= sum_ci w[co][ci] x[ci][p] in the same fp16x3 arithmetic
 *      (conv1x1_x3.hip): NIN's 1x1 layers (models.py:84-110; backward-data passes the transposed weights) and the Gram
 *      backward D x (F - mean) of loss.py:91.  Both operands are plain fp32; the kernel scales and splits them per
 *      workgroup and 32-channel chunk, so weights that change every call need no packing.  x: [n][cin][hw],
 *      w_rowmajor: [cout][cin], y: [n][cout][hw]; bias / out_relu_mask nullable; x_channel_shift (nullable, [cin]) is
 *      subtracted from x first (the Gram's centring).  maua_gram_bwd routes here unless MAUA_GRAM_BWD_X3=0. ---- */
size_t maua_conv1x1_x3_workspace_bytes(int n, int cin, int64_t hw, int cout);
int maua_conv1x1_x3(const float* x, const float* x_channel_shift, const float* w_rowmajor, const float* bias,
                    const float* out_relu_mask, float* y, int n, int cin, int64_t hw, int cout, int relu, int accumulate,
                    void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- ReLU on its own (module path; the engine fuses it into the convs): models.py:130 ------------- */
int maua_relu_fwd(float* x_inplace, int64_t count, maua_stream_t stream);
int maua_relu_bwd(const float* gy, const float* y, float* gx, int64_t count, maua_stream_t stream);

/* ---- pooling: nn.MaxPool2d / nn.AvgPool2d, models.py:119-123 (2x2 s2), :77-80 (3x3 s2 ceil) ------- */
int maua_pool_out_size(int in, int k, int stride, int ceil_mode);
/* mode: 0 = max (first maximum in scan order wins, NaN propagates), 1 = avg (divisor = window clipped to the input). */
int maua_pool2d_fwd(const float* x, float* y, int n, int c, int h, int w, int k, int stride, int ceil_mode, int mode,
                    maua_stream_t stream);
/* gx[n][c][h][w] (overwritten): gather form, recomputes each window's argmax from x (no index tensor, no atomics).
 * relu_mask_by_x != 0: gx is additionally zeroed where x <= 0 (x is a ReLU output; its threshold_backward fused here). */
int maua_pool2d_bwd(const float* gy, const float* x, float* gx, int n, int c, int h, int w, int k, int stride,
                    int ceil_mode, int mode, int relu_mask_by_x, maua_stream_t stream);

/* 2x2 stride-2 max pooling on even planes (every VGG pool, `nn.MaxPool2d(2, 2)` models.py:120) with the decision kept: the
 * forward pass also writes one byte per window (bits 1:0 = position of the first maximum in scan order, ATen's tie rule; bit 2
 * = the winning value is <= 0), the backward pass routes the gradient from those bytes without reading the input map again;
 * `relu_mask` as in maua_pool2d_bwd's relu_mask_by_x.  Bit-identical to maua_pool2d_fwd / maua_pool2d_bwd.
 * The bytes are laid out [image][c / 8][pooled pixel][c % 8] (n * c * h / 2 * w / 2 bytes, c % 8 == 0): the eight channels a staging
 * item of maua_conv3x3_x3w_unpool consumes are one 8-byte load. */
int maua_pool2x2_codes_supported(int n, int c, int h, int w);
int maua_pool2x2_fwd_codes(const float* x, float* y, unsigned char* codes, int n, int c, int h, int w, maua_stream_t stream);
int maua_pool2x2_bwd_codes(const float* gy, const unsigned char* codes, float* gx, int n, int c, int h, int w, int relu_mask,
                           maua_stream_t stream);

/* ---- Gram / covariance matrix: loss.GramMatrix.forward, loss.py:67-91 (torch.mm at :91) ----------- */
/* gram[C][C] = scale * Fc Fc^T with Fc = f[C][hw] (minus row means when `center` != 0, loss.py:87-89).
 * row_mean_out (nullable unless center): receives the C row means.  Split-K over hw with a fixed-order reduction.
 * workspace: maua_gram_workspace_bytes(C, hw) bytes.
 * maua_gram_block: edge of the blocks the split-K kernel of this shape multiplies in - 128 for 128+ channels and 1024+ pixels in whole
 * 64-pixel stages, else 64 (MAUA_GRAM_T128, read once per process: 0 = always 64, 2 = ragged maps too; see gram.hip). */
size_t maua_gram_workspace_bytes(int c, int64_t hw);
int maua_gram_block(int c, int64_t hw);
int maua_gram_fwd(const float* f, float* gram, float* row_mean_out, int c, int64_t hw, float scale, int center,
                  void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- MSE forward + backward in one pass: nn.MSELoss at loss.py:56 (content) and :154/:178 (style) - */
/* loss_out[0] = loss_scale * sum((x - target)^2) ; grad (nullable) (+)= grad_scale * (x - target).
 * mask_grad_by_x != 0: the (accumulated) gradient is zeroed where x <= 0 (x is a ReLU output).
 * Used for ContentLoss on feature maps (grad accumulates into the feature gradient) and for StyleLoss on C x C
 * Gram matrices (grad = the matrix D fed to maua_gram_bwd).  workspace: maua_reduce_workspace_bytes(count). */
size_t maua_reduce_workspace_bytes(int64_t count);
int maua_mse_fwd_bwd(const float* x, const float* target, float* grad, int64_t count, float loss_scale, float grad_scale,
                     int accumulate, int mask_grad_by_x, float* loss_out, void* workspace, size_t workspace_bytes,
                     maua_stream_t stream);
/* The temporal ContentLoss with a reliability mask (loss.py:52-56 `input * self.weights`, set by
 * optim.set_temporal_targets optim.py:35-47): loss_out[0] = loss_scale * sum((x*w - target)^2),
 * grad (nullable) (+)= grad_scale * w * (x*w - target).  x, target, grad: [planes][plane]; weights: [weight_planes][plane]
 * with weight_planes = 1 (one mask for every channel) or = planes. */
int maua_mse_weighted_fwd_bwd(const float* x, const float* weights, const float* target, float* grad, int64_t planes,
                              int64_t plane, int weight_planes, float loss_scale, float grad_scale, int accumulate,
                              float* loss_out, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* Backward of the Gram loss into the feature map: gf[C][hw] (+)= D[C][C] (symmetric) * (f - mean) ;
 * autograd of torch.mm(x, x.t()) at loss.py:91 with the MSE gradient D (scaled by the caller).  relu_mask (nullable,
 * shape of gf): result zeroed where it is <= 0.  Runs as maua_conv1x1_x3 with x_channel_shift = row_mean (fp16x3
 * arithmetic; workspace of maua_gram_workspace_bytes lets small maps split the channel loop); MAUA_GRAM_BWD_X3=0 keeps
 * the fp32-MFMA kernel. */
int maua_gram_bwd(const float* d_sym, const float* f, const float* row_mean, const float* relu_mask, float* gf, int c,
                  int64_t hw, int accumulate, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- total variation: loss.TVLoss.forward, loss.py:224-233, and its autograd ---------------------- */
/* loss_out[0] = strength * (sum|dy| + sum|dx|); grad (+)= strength * d/dx.  workspace as for the MSE. */
int maua_tv_fwd_bwd(const float* x, float* grad, int n, int c, int h, int w, float strength, int accumulate,
                    float* loss_out, void* workspace, size_t workspace_bytes, maua_stream_t stream);

/* ---- deferred loss finishing ------------------------------------------------------------------------
 * The reference adds up its modules' losses after the forward pass (`total_loss += mod.loss`, optim.py:207-211).  The entry
 * points above finish each loss with a launch of its own; these variants leave their per-workgroup partial sums in a "ledger"
 * (caller-allocated, maua_loss_ledger_bytes(frames, slots), zero-filled once: `frames` x `slots` records) and ONE launch of
 * maua_loss_ledger_sum per evaluation turns every filled record into its loss value - the partial sums added in the same fixed
 * order as the immediate entry points, so the values are bit-identical - marks the records empty again and writes, per frame,
 * totals[f] = sum of losses[f][0..slots), left to right.  Slots whose record is empty keep the value an immediate entry point
 * wrote there.  `ledger` arguments of the *_ledger functions point at a frame's first record, `slot` selects the record.
 * maua_gram_fwd_mse_ledger = maua_gram_fwd followed by maua_mse_fwd_bwd of the Gram matrix against `target` (StyleLoss,
 * loss.py:141-157): gram, dmat = grad_scale (G - target) and the loss partials from one finishing launch; it needs
 * maua_gram_mse_ledger_supported(c) (c <= 1408 channels). */
size_t maua_loss_ledger_bytes(int frames, int slots);
int maua_mse_fwd_bwd_ledger(const float* x, const float* target, float* grad, int64_t count, float loss_scale, float grad_scale,
                            int accumulate, int mask_grad_by_x, double* ledger, int slot, maua_stream_t stream);
int maua_tv_fwd_bwd_ledger(const float* x, float* grad, int n, int c, int h, int w, float strength, int accumulate,
                           double* ledger, int slot, maua_stream_t stream);
int maua_gram_mse_ledger_supported(int c);
int maua_gram_fwd_mse_ledger(const float* f, float* gram, float* row_mean_out, int c, int64_t hw, float scale, int center,
                             const float* target, float* dmat, float loss_scale, float grad_scale, double* ledger, int slot,
                             void* workspace, size_t workspace_bytes, maua_stream_t stream);
/* The same chain in two calls, so that ONE finishing launch serves every style layer of an evaluation (their D matrices are needed only
 * when the backward pass starts): maua_gram_partial leaves the split-K slabs of F F^T (and the row means, `center`) in the layer's OWN
 * workspace (maua_gram_workspace_bytes, untouched until the batch call); maua_gram_finish_mse_batch (count <= 8 layers, host arrays of
 * `count` entries) then does for each layer what maua_gram_fwd_mse_ledger's finishing launch does - gram, dmat, the ledger record of
 * `slots[i]` in `ledgers[i]` - with the same arithmetic in the same order: bit-identical results. */
int maua_gram_partial(const float* f, float* row_mean_out, int c, int64_t hw, int center, void* workspace, size_t workspace_bytes,
                      maua_stream_t stream);
/* maua_gram_partial for up to 8 layers at once - at most three partial launches (the layers of one 64-channel tile; the others; those of 128 x 128 blocks)
 * and one first-level fold instead of one to two launches per layer; every layer keeps the plan of its own call: the same slabs, bit for
 * bit.  Host arrays of `count` entries.  slab_counts (nullable; also the last array of maua_gram_finish_mse_batch): an entry > 0 says that
 * the layer's workspace already holds that many 64 x 64 slabs (C <= 64: maua_conv3x3_image_gram) - no partial launch for it, only the
 * fold, and the finishing launch adds that many. */
int maua_gram_partial_batch(int count, const float* const* fs, float* const* means, const int* cs, const int64_t* hws,
                            void* const* workspaces, const size_t* workspace_bytes, const int* slab_counts, maua_stream_t stream);
/* `means` of maua_gram_partial_batch (nullable; entries nullable): a non-null entry makes the layer a covariance-form layer (loss.py:87-89) -
 * the call first computes the row means of all such layers into those arrays (two launches for all of them), then centres their products.
 * maua_gram_row_means: the same means on their own (maua_gram_partial(center = 1) = this + the partial kernel).  workspace: c * 64 doubles
 * (the start of the layer's maua_gram_workspace_bytes buffer will do: the slabs overwrite it later). */
int maua_gram_row_means(const float* f, float* row_mean_out, int c, int64_t hw, void* workspace, size_t workspace_bytes, maua_stream_t stream);
int maua_gram_finish_mse_batch(int count, const void* const* workspaces, float* const* grams, const float* const* targets,
                               float* const* dmats, const int* cs, const int64_t* hws, const float* scales, const float* loss_scales,
                               const float* grad_scales, double* const* ledgers, const int* slots, const int* slab_counts,
                               maua_stream_t stream);
int maua_loss_ledger_sum(double* ledger, int frames, int slots, float* losses, float* totals, maua_stream_t stream);
/* The same launch, which also leaves every filled record's loss BEFORE its rounding to fp32 in losses_f64[frames][slots]
 * (`mod.loss` of optim.py:207-211 as a double; slots whose record is empty are not written): full-size derivative tests
 * difference loss values that agree in their first five digits. */
int maua_loss_ledger_sum_f64(double* ledger, int frames, int slots, float* losses, float* totals, double* losses_f64,
                             maua_stream_t stream);

/* ---- small vector helpers used by the host engine -------------------------------------------------- */
int maua_fill(float* x, int64_t count, float value, maua_stream_t stream);
/* out[n][c][r qy + ry][r qx + rx] (=, or += with `accumulate`) in[n][(ry r + rx) c_out + c][qy][qx] for the h x w pixels of `out`; pixels whose
 * site (qy, qx) lies beyond qh x qw get 0.  The tail of a strided layer's backward-data pass computed as a stride-1 convolution over the
 * output sites: the backward pass of `nn.Conv2d(3, 96, (11, 11), (4, 4))` (NIN's conv1, /root/reference/models.py:84) is a 3x3 convolution
 * 96 -> 48 over the 254 x 254 gradient (taps = the 11 x 11 filter in steps of 4) followed by this (models.conv_strided_bwd_as_3x3). */
/* The image layer's backward-data pass (64 gradient channels -> 1-3 pixel channels; autograd's conv_backward of nn.Conv2d(3, 64, 3,
 * padding=1), /root/reference/models.py:129-130) on the matrix cores (conv_few_mfma.hip, round 5): per gradient pixel the 27 (tap, channel)
 * sums over the 64 channels as one bf16x6 product block, then nine values gathered per output pixel and channel.
 * maua_conv_pack_filters_few_mfma: OIHW weights (64, cin, 3, 3) -> the bank (maua_conv_few_mfma_bank_bytes() bytes; once per weight set).
 * maua_conv3x3_few_mfma: gy (n, 64, h, w) -> gx (n, cin, h, w), written whole; `tile`: 0 = the library's choice, 1 / 2 / 3 = 4 / 8 / 14
 * output rows x 62 columns per workgroup.
 * maua_conv_few_mfma_supported: geometry check (64 filters, 1-3 channels, padding 1, planes whose 64 gradient maps stay below 2 GiB). */
size_t maua_conv_few_mfma_bank_bytes(void);
int maua_conv_pack_filters_few_mfma(const float* w_oihw, void* bank, int cout, int cin, maua_stream_t stream);
int maua_conv_few_mfma_supported(int n, int cin, int h, int w, int cout, int pad);
int maua_conv3x3_few_mfma(const float* gy, const void* bank, float* gx, int n, int cin, int h, int w, int cout, int tile, maua_stream_t stream);

/* The inverse regrouping: out[n][(ry r + rx) c_in + c][qy][qx] = in[n][c][r qy + ry][r qx + rx], 0 for pixels beyond h x w - the head of the
 * same layer's FORWARD pass as a 3x3 stride-1 convolution 48 -> 96 over 256 x 256 sites (models.conv_strided_fwd_as_3x3, conv_x3w.hip). */
int maua_space_to_depth(const float* in, float* out, int n, int c_in, int r, int h, int w, int qh, int qw, maua_stream_t stream);
int maua_depth_to_space(const float* in, float* out, int n, int c_out, int r, int qh, int qw, int h, int w, int accumulate, maua_stream_t stream);
int maua_axpy(float* y, const float* x, float alpha, int64_t count, maua_stream_t stream); /* y += alpha x */
/* out[0] = sum_i in[i] over `count` device floats, fixed order (sums the per-module loss slots, optim.py:207-211). */
int maua_sum_small(const float* in, int count, float* out, maua_stream_t stream);

/* ---- Adam pixel update: torch.optim.Adam([pastiche], lr) as built at optim.py:192-196 ------------- */
/* One step of the single-tensor update (betas 0.9/0.999, eps 1e-8 by default); `step` is 1-based. */
int maua_adam_step(float* x, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, int step, float lr,
                   float beta1, float beta2, float eps, maua_stream_t stream);

/* ---- image-space steps between two optimisation runs ------------------ */
/* utils.match_histogram (reference utils.py:88-151, called at style.py:24,67,71,292): PCA colour transfer.
 * Layout: the reference views the (B,3,H,W) batch as (B,W,H,3) and reshapes the centred tensor with
 * `h.permute(0, 3, 1, 2).reshape(C, -1)` (utils.py:91): the 3 B planes (p = 3 b + c) become 3 rows of B whole planes, so
 * component k of pseudo-pixel (slot j, h, w) is plane k * frames + j (for one image: channel k).  That is reproduced as is.
 * maua_channel_stats: replaces get_histogram's `tensor.mean(...)` and `th.mm(h, h.T)` (utils.py:88-93) for plane slot `slot`
 *   of x [frames][3][h][w], jittered as the reference does (`frame + 1e-3 * th.randn(size=frame.shape)`, utils.py:123-124):
 *   noise_bwhc is that draw in ITS layout [frames][w][h][3], NULL = no jitter.  stats9 (device, doubles) = sum of component
 *   k (3), then sum of products of components (0,0) (0,1) (0,2) (1,1) (1,2) (2,2); fp64 partials per tile in `workspace`,
 *   summed in tile order (deterministic).  One call per slot; the caller lays the results out as [frames][9].
 * maua_color_match_solve: replaces symeig + sqrt + th.inverse + the two th.mm of utils.py:127-139 on the device (one thread,
 *   fp64): mean per true channel, covariance of the 3 rows + eps I, Q = V sqrt(max(L, 0)) V^T for target and (single-frame)
 *   source, coef16 = [M = Qs Qt^-1 (9, row major), mean_target (3), mean_source (3), ok].  ok = 0 where the reference's
 *   `except RuntimeError` path would fire (non-finite statistics, singular Qt).
 * maua_color_match_apply: replaces `ts = M t; match = ts + mu_s; output += match / len(sources)` (utils.py:137-146) for one
 *   plane slot: out (+)= weight * (M (x + amp * noise - mean_target) + mean_source); if any of the n_coef entries of
 *   all_coef has ok == 0 the untouched x is written instead (the reference returns its backup copy for the whole call). */
size_t maua_channel_stats_workspace_bytes(int h, int w);
int maua_channel_stats(const float* x_bchw, const float* noise_bwhc, float noise_amp, int frames, int slot, int h, int w,
                       double* stats9, void* workspace, size_t workspace_bytes, maua_stream_t stream);
int maua_color_match_solve(const double* stats_target, int frames, int64_t pixels_target, const double* stats_source,
                           int64_t pixels_source, float eps, float* coef16, maua_stream_t stream);
int maua_color_match_apply(const float* x_bchw, const float* noise_bwhc, float noise_amp, const float* coef16, const float* all_coef,
                           int n_coef, float weight, int accumulate, int frames, int slot, int h, int w, float* out_bchw,
                           maua_stream_t stream);
/* F.interpolate(..., mode="bilinear", align_corners=False) as style.py:38-66 calls it (content / style / pastiche rescaling
 * between scales): y[planes][oh][ow] from x[planes][h][w]; source index = max(0, scale * (dst + 0.5) - 0.5) in fp32 with
 * scale = float(1 / scale_factor) (scale_factor form) or in / out (size form) - ATen's area_pixel_compute_source_index. */
int maua_resize_bilinear(const float* x, float* y, int planes, int h, int w, int oh, int ow, float scale_h, float scale_w,
                         maua_stream_t stream);
/* load.deprocess (reference load.py:47-52) + ToPILImage's truncation: BGR mean-subtracted [3][h][w] -> RGB uint8 [h][w][3],
 * byte(clamp((x + mean) / 255, 0, 1) * 255) with the same fp32 operations. */
int maua_deprocess_u8(const float* x_bgr_chw, unsigned char* out_rgb_hwc, int h, int w, float mean_b, float mean_g, float mean_r,
                      maua_stream_t stream);

/* ---- batches of independent frames (vid_img without optical flow, reference style.py:192-290) ---------- */
/* The library's tuning constants - routing thresholds and experiment switches (which kernel family a shape gets, forced K splits, the
 * 128 x 128 Gram blocks ...): the host side's planner configuration (maua-style_amd/plan.py lists every name with its default and
 * meaning) sets them when it loads the library; csrc/ reads them at every use and never reads the environment.  The reference has no
 * counterpart (its knobs are argparse flags, /root/reference/config.py); this replaces the per-file `getenv` calls of rounds 1-4.
 * maua_set_tuning: MAUA_E_INVAL for a name the library does not know.  maua_get_tuning: the value set, or `dflt`. */
int maua_set_tuning(const char* name, double value);
double maua_get_tuning(const char* name, double dflt);

/* The convolution entry points take a batch dimension n; deterministic split-K sums its slabs in a fixed order, but HOW MANY
 * slabs a layer is cut into is a cost-model decision that would depend on n.  To keep a frame's result independent of how
 * many frames happen to share a launch, the cost models count the frames the CALLER PLANS per launch: the host sets that
 * number once per job (style.vid_img: frames_per_batch(size); 1 = single images, the default) and every launch of the job -
 * a full batch, the short last batch, a single frame - uses the same split.  A setting of the calling host thread (round 4; it was
 * process-wide before), read at launch time and by the *_workspace_bytes functions on that thread (set it before sizing
 * workspaces): concurrent jobs with different plans run on a thread each. */
void maua_set_split_batch_hint(int frames);
int maua_get_split_batch_hint(void);

/* Split channel loops finished INSIDE the producing launch (round 6).  A launch that splits its channel loop k ways leaves k slabs of partial
 * sums and - until now always - a second launch adds them (conv_splitk_finish_kernel: bias, ReLU, mask, pool; 7-12 us for kilobytes to
 * megabytes of work: 19 such launches were 9 % of a 512 x 512 iteration).  With ARRIVAL COUNTERS the producing launch does it itself: the
 * workgroup that arrives last at its tile adds the other splits' slabs to its own sums - in split order, the additions of the second
 * launch, bit for bit - and runs the one-pass epilogue; no workgroup waits for another (cdna_hip_programming.md: in-launch split-K
 * reduction, write-through form).  The reference has no counterpart (one `F.conv2d` per layer, /root/reference/models.py:129-130).
 *   maua_conv_arm_workspace(workspace, counters, counter_bytes, zero, stream): `counters` = 16 KiB (4096 words, 16-byte aligned) of device
 *     memory that the library zeroes here (on `stream`; `zero` = 0: the caller vouches that they are zero already - pointing the thread
 *     back at a workspace it armed before, e.g. while a stream is being captured) and that every in-launch finish leaves zeroed again.  From then on the launches
 *     of the CALLING HOST THREAD that are handed exactly this `workspace` pointer finish splits of at most `finish_in_launch_max_ks`
 *     (tuning constant, default 4) slabs in the launch: maua_conv3x3_x3q / _relu_pool / _unpool and maua_conv3x3_x3w / _relu_pool /
 *     _unpool / _gram.  Any other workspace, larger splits and the other kernel families keep the second launch.  Results are
 *     bit-identical either way.  The counters belong to the workspace: one stream of launches at a time, like its slabs.  NULL
 *     workspace or counters: disarm.  Per host thread, like the batch hint.
 * The *_workspace_bytes functions already return enough for either form. */
int maua_conv_arm_workspace(void* workspace, void* counters, size_t counter_bytes, int zero, maua_stream_t stream);

/* ---- L-BFGS pixel update: torch.optim.LBFGS as configured at optim.py:180-191 --------------------- */
/* Device-resident state: `state` is an opaque caller-allocated buffer of maua_lbfgs_state_bytes(count, history)
 * bytes holding the (s, y) history slab [2*history][count], the previous gradient, the direction, the Gram matrix of
 * the history and all scalars.  maua_lbfgs_init zeroes the bookkeeping (not the slab).
 * maua_lbfgs_iterate consumes the gradient (and, optionally, the loss value: a device scalar, NULL to skip the loss test)
 * of the current x and moves x, replicating one trip of the loop in LBFGS.step without line search: first call d = -g,
 * t = min(1, 1/|g|_1) * lr; later calls push the curvature pair when y.s > 1e-10, run the two-loop recursion (evaluated
 * in the coefficient space of the stored vectors), t = lr.  The stop tests of lbfgs.py are all evaluated on the device
 * (x left untouched, status flag raised): max|g| <= tolerance_grad, g.d > -tolerance_change, max|t d| <= tolerance_change
 * of the previous move, |loss - previous loss| < tolerance_change.  history <= 254.  No host synchronisation.
 * maua_lbfgs_status copies {n_iter, history_len, stopped, last g.d, last t} (5 floats) to a device buffer. */
size_t maua_lbfgs_state_bytes(int64_t count, int history);
int maua_lbfgs_init(void* state, size_t state_bytes, int64_t count, int history, maua_stream_t stream);
int maua_lbfgs_iterate(void* state, float* x, const float* grad, const float* loss, int64_t count, int history, float lr,
                       float tolerance_change, float tolerance_grad, maua_stream_t stream);
int maua_lbfgs_status(const void* state, int64_t count, int history, float* out5, maua_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAUA_HIP_H */
