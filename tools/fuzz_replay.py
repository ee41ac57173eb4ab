#!/usr/bin/env python3
"""Replay one fuzz_parity.py case (the JSON after '::') on the GPU box and print where HIP and oracle differ."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np, torch
import ce_oracle as O
import fuzz_cases as F
from srsran_ce_pytorch_amd import estimator as E
rec = json.loads(sys.argv[1])
case, extras = rec["case"], rec["extras"]
b = F.realize(case, extras)
g = torch.as_tensor(b.grids, device="cuda:0")[None]
if not extras["layout_ref"]:
    g = g.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
out = E.estimate(g, torch.as_tensor(b.pilots, device="cuda:0"), b.beta, b.hop1, b.hop2, b.config, interp=extras["interp"])
for it in range(2):
    st = []
    ref = O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config, interp=extras["interp"], stages=st)
    ch = out[0][0, it].cpu().numpy()
    d = np.abs(ch - ref[0])
    sc, sym, l = np.unravel_index(d.argmax(), d.shape)
    print(f"item {it}: max|d| {d.max():.3e} of max|h| {np.abs(ref[0]).max():.3f} at sc {sc} sym {sym} layer {l}; per-symbol max {np.round(d.max(axis=(0, 2)) * 1e6, 2)} e-6")
    print("   scalars HIP   ", [float(o[0, it]) if o.numel() else None for o in out[1:]])
    print("   scalars oracle", [float(x) if x is not None else None for x in ref[1:]])
    print("   oracle TA bins / powers", [(s["ta_bin"], s["ta_pw"]) for s in st])
