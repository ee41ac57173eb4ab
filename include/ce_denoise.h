/*
 * ce_denoise.h -- EXTENSION of libce_hip.so with NO counterpart in the reference ("parity unpinned").
 *
 * BASELINE.json's north_star / configs[4] name a "CNN denoiser (Conv2d over the RE grid), fp16, MFMA path".  The
 * reference (pjookim/srsran-ce-pytorch) has no learned network: src/ce_dl_cnn.py is a fixed 3-tap in-painting
 * (built as CE_INTERP_CNN in ce_hip.h, parity-pinned).  This header adds the Conv2d operator the north_star asks
 * for as an opt-in post-processor of the channel-estimate grid; its only oracle is the build's own numpy
 * restatement, oracle/ce_denoise_oracle.py.  Weights come from the caller (nothing is shipped or downloaded).
 *
 * Operator, per (slot, port, layer) plane h[subcarrier][symbol] (complex64, 14 symbols):
 *     x0 = (Re h, Im h)                                            2 channels, rounded to fp16
 *     x1 = ReLU(conv3x3(x0, W1) + b1)                              16 channels, fp16
 *     x2 = ReLU(conv3x3(x1, W2) + b2)                              16 channels, fp16
 *     out = h + (conv3x3(x2, W3) + b3)                             residual, float32
 * conv3x3 = cross-correlation with zero padding ("same") over (subcarrier, symbol); fp16 operands, float32
 * accumulation on v_mfma_f32_16x16x32_f16.
 */
#ifndef CE_DENOISE_H
#define CE_DENOISE_H

#include "ce_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CE_DN_CHANNELS 16
#define CE_DN_SYMBOLS 14

typedef struct ce_denoiser ce_denoiser; /* opaque: fp16 weights in MFMA fragment order on the device */

/* Weights are host float32 in PyTorch Conv2d order [c_out][c_in][k_subcarrier][k_symbol]:
 * w1 [16][2][3][3], b1 [16], w2 [16][16][3][3], b2 [16], w3 [2][16][3][3], b3 [2].  Synchronous. */
int ce_denoiser_create(int32_t device, const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* w3, const float* b3, ce_denoiser** out);
void ce_denoiser_destroy(ce_denoiser* dn);

/* In place on a dense complex64 [n_items][n_sc][14][n_layers] buffer (ce_estimate_batch's ch_est layout), one
 * workgroup per (item, layer) plane, asynchronous on `stream`. */
int ce_denoise_batch(const ce_denoiser* dn, void* ch_est, int64_t n_items, int32_t n_sc, int32_t n_sym,
                     int32_t n_layers, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CE_DENOISE_H */
