#!/usr/bin/env python3
"""Batch-shape fuzzer (GPU box): the same slots through ONE launch of [B slots x R ports] with per-slot pilots, and slot by slot / port by
port through separate launches -- the results must be the same bits.  What it exercises is everything between the boundary's
(slot, port) indices and a work item: the XCD-aware placement of a slot's ports (ragged tails), the four-items-per-workgroup packing of
the wave-per-item kernel (dead waves in the last workgroup), per-slot pilot strides.  No oracle involved: HIP against HIP.

    python tools/fuzz_batch_shapes.py [--n 300] [--seed 0] [--max-grid 52] [--wide]"""
import argparse, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import fuzz_cases as F
from srsran_ce_pytorch_amd import estimator as E, synth as S


def bits(t):
    return torch.view_as_real(t) if t.is_complex() else t.view(torch.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-grid", type=int, default=52)
    ap.add_argument("--wide", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    bad = skipped = 0
    shapes = set()
    for i in range(a.n):
        rng = np.random.default_rng([a.seed, i])
        case, extras = F.draw(rng, a.max_grid, a.wide)
        if case["smoothing"] == "mmse":
            case["smoothing"] = "filter"
        B, R = int(rng.integers(1, 7)), int(rng.integers(1, 10))
        try:
            slots = [F.realize(dict(case, seed=case["seed"] + 17 * k), extras, R) for k in range(B)]
            rx = torch.as_tensor(np.stack([s.grids for s in slots]), device=dev)                       # [B, R, n_sc, n_sym]
            if not extras["layout_ref"]:
                rx = rx.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
            pil = torch.as_tensor(np.stack([s.pilots for s in slots]), device=dev)                      # [B, n_re, n_dmrs, L]: per-slot pilots
            b0 = slots[0]
            whole = E.estimate(rx, pil, b0.beta, b0.hop1, b0.hop2, b0.config, interp=extras["interp"])
            torch.cuda.synchronize()
        except (ValueError, AssertionError, NotImplementedError):
            skipped += 1
            continue
        shapes.add((B, R))
        ok = True
        for k in range(B):                                                                               # slot by slot ...
            one = E.estimate(rx[k:k + 1], pil[k], b0.beta, b0.hop1, b0.hop2, b0.config, interp=extras["interp"])
            ok = ok and all(torch.equal(bits(w[k:k + 1]), bits(o)) for w, o in zip(whole, one) if w.numel() and o.numel())
        k, r = int(rng.integers(B)), int(rng.integers(R))                                                # ... and one port alone
        one = E.estimate(rx[k:k + 1, r:r + 1], pil[k], b0.beta, b0.hop1, b0.hop2, b0.config, interp=extras["interp"])
        ok = ok and all(torch.equal(bits(w[k:k + 1, r:r + 1]), bits(o)) for w, o in zip(whole, one) if w.numel() and o.numel())
        if not ok:
            bad += 1
            print(f"[{i}] B={B} R={R}: the batched launch and the separate launches differ :: {case} {extras}", flush=True)
        if (i + 1) % 100 == 0:
            print(f"... {i + 1} cases, {bad} differences, {skipped} skipped (inputs the estimator refuses, as the reference does)", flush=True)
    print(f"done: {a.n} cases over {len(shapes)} distinct (slots, ports) shapes, {bad} differences, {skipped} skipped")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
