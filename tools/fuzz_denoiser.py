#!/usr/bin/env python3
"""Dev tool (GPU box): random shapes / weights / magnitudes through the Conv2d denoiser extension vs its numpy oracle."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
import numpy as np, torch
import ce_denoise_oracle as DO
from srsran_ce_pytorch_amd.denoiser import Denoiser, random_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for i in range(n):
    n_sc = int(rng.choice([1, 2, 3, 12, 31, 32, 33, 63, 64, 65, 95, 96, 97, 300, 624, 1272])) if rng.random() < 0.7 else int(rng.integers(1, 700))
    L, items = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    w = random_weights(int(rng.integers(1 << 30)), gain=float(rng.uniform(0.2, 1.2)))
    scale = float(10 ** rng.uniform(-2, 1))
    h = (scale * (rng.standard_normal((items, n_sc, 14, L)) + 1j * rng.standard_normal((items, n_sc, 14, L)))).astype(np.complex64)
    want = DO.denoise(h, w)
    t = torch.from_numpy(h.copy()).cuda()
    Denoiser(w)(t)
    err = float(np.abs(t.cpu().numpy() - want).max() / np.abs(h).max())
    worst = max(worst, err)
    flag = "" if err <= 2e-3 else "  <-- MISMATCH"
    print(f"[{i}] n_sc={n_sc} L={L} items={items} scale={scale:.2g}: err {err:.2e} of max|h|, correction {np.abs(want - h).max() / np.abs(h).max():.2f}{flag}", flush=True)
print("worst", worst)
sys.exit(0 if worst <= 2e-3 else 1)
