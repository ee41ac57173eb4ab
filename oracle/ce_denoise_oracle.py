"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the Conv2d denoiser EXTENSION (include/ce_denoise.h).

"Parity unpinned": the reference (pjookim/srsran-ce-pytorch) has no learned denoiser -- its `ce_dl_cnn.py` is a fixed
3-tap in-painting -- so there is nothing upstream to check this against; this file is the only oracle of
`ce_denoise_batch`.  It follows the operator definition in the header literally: fp16-rounded weights and
activations, wide accumulation, zero padding at every layer, float32 residual."""
from __future__ import annotations

import numpy as np


def _h(x: np.ndarray) -> np.ndarray:
    """round to fp16 and back (what the kernel stores between layers)"""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float64)


def conv3x3(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """x [c_in, n_sc, n_sym], w [c_out, c_in, 3, 3] (cross-correlation, zero padding) -> [c_out, n_sc, n_sym], float64."""
    c_in, n_sc, n_sym = x.shape
    xp = np.zeros((c_in, n_sc + 2, n_sym + 2), np.float64)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], n_sc, n_sym), np.float64) + np.asarray(b, np.float64)[:, None, None]
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("oi,iyx->oyx", w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + n_sc, kx:kx + n_sym])
    return out


def denoise_plane(h: np.ndarray, w1, b1, w2, b2, w3, b3) -> np.ndarray:
    """h [n_sc, 14] complex64 -> denoised complex64."""
    x0 = _h(np.stack([h.real, h.imag]))
    x1 = _h(np.maximum(conv3x3(x0, _h(w1), b1), 0.0))
    x2 = _h(np.maximum(conv3x3(x1, _h(w2), b2), 0.0))
    r = conv3x3(x2, _h(w3), b3)
    out = h.astype(np.complex64).copy()
    out.real += r[0].astype(np.float32)
    out.imag += r[1].astype(np.float32)
    return out


def denoise(ch_est: np.ndarray, weights: dict) -> np.ndarray:
    """ch_est [..., n_sc, 14, L] complex64: every (.., layer) plane on its own."""
    ch = np.asarray(ch_est, np.complex64)
    flat = ch.reshape((-1,) + ch.shape[-3:])
    out = np.empty_like(flat)
    for i in range(flat.shape[0]):
        for l in range(flat.shape[-1]):
            out[i, :, :, l] = denoise_plane(flat[i, :, :, l], **weights)
    return out.reshape(ch.shape)
