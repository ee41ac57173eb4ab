"""Conv2d denoiser EXTENSION (include/ce_denoise.h; "parity unpinned": the reference has no learned network).

`Denoiser(weights, device)` packs caller-supplied float32 Conv2d weights into MFMA fragment order on the GPU;
calling it post-processes a channel-estimate batch in place (one fused kernel, fp16 operands, float32 accumulate).
No weights ship with this repo: `random_weights` draws a small-gain set for tests and the bench."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import _lib

SHAPES = {"w1": (16, 2, 3, 3), "b1": (16,), "w2": (16, 16, 3, 3), "b2": (16,), "w3": (2, 16, 3, 3), "b3": (2,)}


def random_weights(seed: int = 0, gain: float = 0.5) -> Dict[str, np.ndarray]:
    """He-style draws scaled by `gain` (the residual branch stays a small correction); float32 numpy arrays."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in SHAPES.items():
        if name.startswith("w"):
            out[name] = (gain * rng.standard_normal(shape) * np.sqrt(2.0 / (shape[1] * 9))).astype(np.float32)
        else:
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
    return out


class Denoiser:
    def __init__(self, weights: Dict[str, np.ndarray], device="cuda:0"):
        self._lib = _lib.load()
        self.device = torch.device(device)
        arrs = []
        for name, shape in SHAPES.items():
            a = np.ascontiguousarray(np.asarray(weights[name], np.float32))
            if a.shape != shape:
                raise ValueError(f"{name}: shape {a.shape}, expected {shape}")
            arrs.append(a)
        self._handle = C.c_void_p()
        ptrs = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs]
        rc = self._lib.ce_denoiser_create(self.device.index or 0, *ptrs, C.byref(self._handle))
        if rc != 0:
            raise RuntimeError(_lib.last_error())

    def __call__(self, ch_est: torch.Tensor) -> torch.Tensor:
        """In place on `[..., n_sc, 14, L]` complex64 (dense, on this denoiser's device); returns `ch_est`."""
        if ch_est.dtype != torch.complex64 or not ch_est.is_contiguous() or ch_est.device != self.device or ch_est.dim() < 3:
            raise ValueError("ch_est must be a dense complex64 [..., n_sc, n_sym, L] tensor on the denoiser's device")
        n_sc, n_sym, L = ch_est.shape[-3:]
        n_items = ch_est.numel() // (n_sc * n_sym * L) if ch_est.numel() else 0
        rc = self._lib.ce_denoise_batch(self._handle, C.c_void_p(ch_est.data_ptr()), n_items, n_sc, n_sym, L,
                                        C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc == _lib.CE_ERR_UNSUPPORTED:
            raise NotImplementedError(_lib.last_error())
        if rc != 0:
            raise ValueError(_lib.last_error()) if rc == _lib.CE_ERR_INVALID else RuntimeError(_lib.last_error())
        return ch_est

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.ce_denoiser_destroy(h)
