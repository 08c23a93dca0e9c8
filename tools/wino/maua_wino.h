/* maua_wino.h - C ABI of the REJECTED Winograd experiment (round 3; profiles/probes_r03.md section 1): parity-green, slower than
 * conv_x3w on every VGG layer.  Not part of libmaua_hip.so since round 4: tools/wino/build.sh builds tools/_build/libmaua_wino.so
 * from conv_wino.hip and the product library's common.hpp; tools/wino/wino.py binds it. */
#ifndef MAUA_WINO_H
#define MAUA_WINO_H
#include "../../include/maua_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- Winograd F(2x2, 3x3) on the fp16x3 split (tools/wino/conv_wino.hip) -----------------------------------
 * The same layer arithmetic as maua_conv3x3_x3w - `nn.Conv2d(cin, c, 3)` + `nn.ReLU(inplace=True)`, models.py:129-130, and
 * its backward-data pass - with 16 products per 2x2 outputs instead of 36: the filters are transformed once per weight
 * version (G g G^T in fp64, split into two fp16 parts under a power-of-two scale kept in the bank's header), the inputs
 * while they are staged (B^T d B in fp32).  maua_conv_pack_filters_wino fills the forward and / or the backward-data bank
 * (either pointer may be null; maua_conv_wino_bank_bytes(cout, cin) / (cin, cout) bytes) without a host synchronisation.
 * maua_conv3x3_wino: y = [mask > 0] relu?(conv(x) + bias? + y?)  for cin % 16 == 0, pad <= 2 (backward-data: pad' = 2 - pad
 * with the backward bank and cout = the layer's cin); out_relu_mask is output-shaped and nullable. */
size_t maua_conv_wino_bank_bytes(int cout, int cin);
int maua_conv_pack_filters_wino(const float* w_oihw, void* bank_fwd, void* bank_bwd, int cout, int cin, maua_stream_t stream);
int maua_conv_wino_supported(int cin, int h, int w, int pad);
int maua_conv3x3_wino(const float* x, const void* bank, const float* bias, const float* out_relu_mask, float* y, int n, int cin,
                      int h, int w, int cout, int pad, int relu, int accumulate, maua_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
