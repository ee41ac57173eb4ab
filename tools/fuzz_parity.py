#!/usr/bin/env python3
"""Differential fuzzer (GPU box): random slot geometries through the HIP estimator and the CPU oracle.

    python tools/fuzz_parity.py [--n 200] [--seed 0] [--max-grid 106]

Draws grid sizes, 1-2 hops (disjoint, identical or partly shared symbol ranges), 1-4 DM-RS symbols per hop, RE
patterns beyond the two NR types, 1-4 layers, every smoothing mode, both interpolators; compares with the test
suite's protocol (tests/conftest.py::check_outputs) and requires the same exception class when the oracle raises.
Prints one line per disagreement with the case as JSON (replay: S.build_case(case, 2)).  Every disagreement found
so far became a case of tests/test_hip_parity.py."""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import numpy as np
import torch
import ce_oracle as O
from conftest import check_outputs
from srsran_ce_pytorch_amd import estimator as E, synth as S

SINGLE = [S.TYPE1_CDM0, S.TYPE1_CDM1, S.TYPE2_CDM0, S.TYPE2_CDM1, [1] * 12, [1, 0, 0, 0] * 3, [0, 0, 1, 0] * 3,
          [1, 0, 0, 0, 0, 0] * 2, [1] + [0] * 11, [1, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0],
          [1, 0, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0], [1, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 0], [1] * 11 + [0]]
PAIRS = [[S.TYPE1_CDM0, S.TYPE1_CDM1], [S.TYPE2_CDM0, S.TYPE2_CDM1], [S.TYPE2_CDM1, [0, 0, 0, 0, 1, 1] * 2],
         [[1, 0, 0, 0] * 3, [0, 1, 0, 0] * 3], [S.TYPE1_CDM1, S.TYPE1_CDM0]]


def draw(rng, max_grid):
    grid = int(rng.choice([g for g in (6, 25, 52, 106, 273) if g <= max_grid]))
    layers = int(rng.choice([1, 1, 1, 2, 3, 4]))
    masks = [SINGLE[rng.integers(len(SINGLE))]] if layers <= 2 else PAIRS[rng.integers(len(PAIRS))]
    n_hops = int(rng.choice([1, 1, 2]))
    n_prbs = int(rng.integers(1, grid + 1)) if rng.random() < 0.3 else int(rng.integers(1, min(grid, 12) + 1))
    interp = "cnn" if rng.random() < (0.15 if grid <= 52 else 0.05) else "linear"   # wide CNN cases: closed-form writer for converged masks
    hops = []
    style = rng.choice(["split", "full", "partial"]) if n_hops == 2 and interp == "linear" else "split"
    for h in range(n_hops):
        lo, hi = (0, 14) if n_hops == 1 else ((0, 7) if h == 0 else (7, 14))
        nd = int(rng.integers(1, 5)) if n_hops == 1 else int(rng.integers(1, 4))
        dm = sorted(rng.choice(np.arange(lo, hi), size=min(nd, hi - lo), replace=False).tolist())
        if n_hops == 1:
            start = int(rng.integers(0, 3)) if rng.random() < 0.3 else 0
            n_alloc = 14 - start - (int(rng.integers(0, 3)) if rng.random() < 0.3 else 0)
        elif style == "split":
            start, n_alloc = lo, hi - lo
        elif style == "full":
            start, n_alloc = 0, 14
        else:
            start, n_alloc = (0, 10) if h == 0 else (5, 9)
        hops.append(S.hop_spec(dm, int(rng.integers(0, grid - n_prbs + 1)), n_prbs, start, n_alloc, masks))
    if n_hops == 2 and rng.random() < 0.5:          # same DM-RS count in both hops (register path)
        k = min(len(hops[0]["dmrs_symbols"]), len(hops[1]["dmrs_symbols"]))
        hops[0]["dmrs_symbols"], hops[1]["dmrs_symbols"] = hops[0]["dmrs_symbols"][:k], hops[1]["dmrs_symbols"][:k]
    n_sym = 14 if (rng.random() < 0.9 or interp == "cnn") else 12          # 12: generic writer; CFO ramp impossible (T:928)
    if n_sym == 12:
        for h in hops:
            h["dmrs_symbols"] = sorted({min(s, 11) for s in h["dmrs_symbols"]})
            h["start_symbol"], h["n_alloc"] = min(h["start_symbol"], 11), min(h["n_alloc"], 12 - min(h["start_symbol"], 11))
        if n_hops == 2 and set(hops[0]["dmrs_symbols"]) & set(hops[1]["dmrs_symbols"]):
            n_sym = 14
    smoothing = str(rng.choice(["none", "mean", "filter", "filter", "mmse"])) if interp == "linear" else str(rng.choice(["none", "mean", "filter"]))
    case = S.case_spec("fuzz", grid, hops, n_layers=layers, smoothing=smoothing, n_sym=n_sym,
                       cfo_compensate=bool(rng.random() < 0.8), scs=float(rng.choice([15e3, 30e3, 60e3])),
                       seed=int(rng.integers(1 << 30)), cfo_hz=float(rng.uniform(-400, 400)), delay_ns=float(rng.uniform(0, 400)))
    return case, interp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-grid", type=int, default=106)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    bad = unsupported = raised = near_ties = illcond = 0
    for i in range(a.n):
        case, interp = draw(rng, a.max_grid)
        tag = json.dumps(dict(case, interp=interp))
        try:
            b = S.build_case(case, 2)
        except Exception as e:                       # generator limits (e.g. mask/pilot shapes), not a finding
            continue
        if interp == "cnn":
            b.config.CNNSmoothingAlpha = float(rng.choice([0.0, 0.4]))
        if case["smoothing"] == "mmse":              # extension: checked against its own oracle
            b.config.MMSEDelaySpread, b.config.MMSENoiseToSignal = float(rng.choice([0.3e-6, 1.2e-6])), float(rng.choice([0.01, 0.1]))
        scattered = rng.random() < 0.15              # maskPRBs not the PRBstart..+nPRBs run: table-lookup paths (pilot positions, TA map)
        if scattered:
            for hop in (b.hop1, b.hop2):
                n = int(getattr(hop, "nPRBs", 0))
                if n:
                    mp = np.zeros(case["n_prb_grid"], bool)
                    mp[rng.choice(case["n_prb_grid"], size=n, replace=False)] = True
                    hop.maskPRBs = mp
        layout_ref = rng.random() < 0.3              # dense [sc][sym] grids (the reference's layout) instead of [sym][sc]
        tag = json.dumps(dict(case, interp=interp, scattered=bool(scattered), layout_ref=bool(layout_ref)))
        want, werr = [], None
        try:
            for it in range(2):
                want.append(O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config, interp=interp))
        except (ValueError, AssertionError, IndexError) as e:   # IndexError: a DM-RS mask shorter than the 14 symbol start times (T:440-447)
            werr = e
        try:
            g = torch.as_tensor(b.grids, device=dev)[None]
            if not layout_ref:
                g = g.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
            out = E.estimate(g, torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config, interp=interp)
            torch.cuda.synchronize()
        except NotImplementedError as e:
            unsupported += 1
            print(f"[{i}] UNSUPPORTED {e} :: {tag}", flush=True)
            continue
        except (ValueError, AssertionError) as e:
            if isinstance(werr, IndexError) and isinstance(e, ValueError):
                raised += 1                          # the reference crashes on the shape; the boundary reports it as invalid
            elif werr is None or type(e) is not type(werr):
                bad += 1
                print(f"[{i}] HIP raised {type(e).__name__}: {e}; oracle: {werr!r} :: {tag}", flush=True)
            else:
                raised += 1
            continue
        if werr is not None:
            bad += 1
            print(f"[{i}] oracle raised {werr!r}, HIP did not :: {tag}", flush=True)
            continue
        ch = out[0][0].cpu().numpy()
        sc = [t[0].cpu().numpy() if t.numel() else None for t in out[1:]]
        for it in range(2):
            ref = want[it]
            got = [sc[0][it], sc[1][it], sc[2][it], sc[3][it], np.nan if sc[4] is None else sc[4][it]]
            rs = [ref[1], ref[2], ref[3], ref[4], np.nan if ref[5] is None else ref[5]]
            for j in range(5):                       # both non-finite (1 pilot: noise = residual / 0 is inf or nan by rounding) counts as equal
                if not np.isfinite(rs[j]) and not np.isfinite(got[j]) and (j != 4 or np.isnan(rs[j]) == np.isnan(got[j])):
                    rs[j] = got[j] = 0.0 if j != 4 else np.nan
            if b.pilots.shape[0] == 1:
                got[3] = rs[3]                       # a single pilot: every TA bin ties, the arg-max is rounding noise
            elif b.pilots.shape[0] <= 36 and 0.4 / len(case["hops"]) / 4096 / case["scs"] < abs(got[3] - rs[3]) <= 1.001 / 4096 / case["scs"]:
                near_ties += 1                       # <= 36 pilots: the IFFT main lobe spans >= 50 bins, a peak midway between two bins ties to ~1e-7
                got[3] = rs[3]                       # (the slot's TA is the mean over hops: steps of 1 / n_hops bins)
            elif b.pilots.shape[0] <= 2 and got[3] != rs[3]:
                near_ties += 1                       # two pilots: |IFFT| is periodic, several bins tie exactly
                got[3] = rs[3]
            if np.isfinite(rs[4]) and abs(got[4] - rs[4]) <= 5e-8 * case["scs"]:
                got[4] = rs[4]                       # float32 floor of the CFO: |d angle| ~ 3e-7 rad whatever the angle
            if scattered and np.isfinite(rs[4]) and abs(rs[4] - case["cfo_hz"]) > 50.0:
                # the synthetic channel is laid out for the contiguous run, so with a scattered mask the CFO correlation is a
                # sum of incoherent terms: its angle (and the ramp built from it) is ill-conditioned in float32 on BOTH sides
                illcond += 1
                break
            try:
                # "mean" smoothing can cancel to a small band mean: float32 rounding scales with the pilots (|H| ~ 1), not the result
                tol_ch = 2e-5 * max(1.0, 0.7 / float(np.abs(ref[0]).max())) if case["smoothing"] == "mean" else 2e-5
                check_outputs(ch[it], got, ref[0], rs, tol_ch, 2e-5, f"fuzz[{i}][{it}]")
            except AssertionError as e:
                bad += 1
                print(f"[{i}] MISMATCH {e} :: {tag}", flush=True)
                break
        if (i + 1) % 50 == 0:
            print(f"... {i + 1} cases, {bad} disagreements, {unsupported} unsupported, {raised} agreed errors", flush=True)
    print(f"done: {a.n} cases, {bad} disagreements, {unsupported} unsupported, {raised} agreed errors, {near_ties} TA near-ties (one bin, <= 36 pilots), {illcond} skipped (incoherent CFO correlation under a scattered mask)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
