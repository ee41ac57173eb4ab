#!/usr/bin/env python3
"""Dev tool (GPU box): time the Conv2d denoiser extension on a resident 273-PRB batch; agreement with its oracle on one plane."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
import numpy as np, torch
import ce_denoise_oracle as DO
from srsran_ce_pytorch_amd.denoiser import Denoiser, random_weights
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = random_weights(3)
dn = Denoiser(w)
t = (0.7 * torch.randn(slots, 4, 3276, 14, 1, dtype=torch.complex64, device="cuda:0"))
ref_in = t[0, 0].cpu().numpy()
dn(t); torch.cuda.synchronize()
err = np.abs(t[0, 0].cpu().numpy() - DO.denoise(ref_in, w)).max()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(3): dn(t)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 3)
planes = slots * 4
flops = planes * 3276 * 14 * 2 * (2 * 16 * 9 + 16 * 16 * 9 + 16 * 2 * 9)
issued = planes * (3276 / 32) * (36 * 1 + 34 * 5 + 8 * 9) * 16 * 16 * 32 * 2   # MFMAs per 32-row strip: layer 1, layer 2, layer 3 (banded 4-row blocks)
print(f"{slots} slots x 4: {best:.3f} ms  useful {flops / best / 1e9:.1f} TFLOP/s  issued-MFMA {issued / best / 1e9:.1f} TFLOP/s  "
      f"bytes {planes * 3276 * 14 * 16 / best / 1e6:.0f} GB/s  max|err| vs oracle {err:.2e}")
