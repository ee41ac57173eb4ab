"""Seeded random slot geometries for the differential tests (tests/test_hip_fuzz.py on the GPU box,
tools/fuzz_parity.py for long runs): grid sizes 6-273 PRB, 1-2 hops with disjoint / identical / partly shared symbol
ranges, 1-4 DM-RS symbols per hop, RE patterns beyond the two NR types, 1-4 layers, every smoothing mode, both
interpolators, 12- and 14-symbol grids, contiguous and scattered PRB masks (the received pilots are laid out for the
mask the estimator is given, so the CFO correlation is coherent), both input layouts."""
from __future__ import annotations

import numpy as np

from conftest import TA_TIE_RATIO, check_outputs, ta_alternatives_from
from srsran_ce_pytorch_amd import synth as S

SINGLE = [S.TYPE1_CDM0, S.TYPE1_CDM1, S.TYPE2_CDM0, S.TYPE2_CDM1, [1] * 12, [1, 0, 0, 0] * 3, [0, 0, 1, 0] * 3,
          [1, 0, 0, 0, 0, 0] * 2, [1] + [0] * 11, [1, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0],
          [1, 0, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0], [1, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 0], [1] * 11 + [0]]
PAIRS = [[S.TYPE1_CDM0, S.TYPE1_CDM1], [S.TYPE2_CDM0, S.TYPE2_CDM1], [S.TYPE2_CDM1, [0, 0, 0, 0, 1, 1] * 2],
         [[1, 0, 0, 0] * 3, [0, 1, 0, 0] * 3], [S.TYPE1_CDM1, S.TYPE1_CDM0]]


def draw(rng, max_grid=273, wide=False):
    """One random case: returns ``(case, extras)``; ``extras`` = interp, layout_ref, cnn_alpha, mmse parameters.
    ``wide`` (tools/fuzz_parity.py --wide; never the suite's pinned slice, whose draws must not move): also what no NR
    configuration has but the reference accepts -- up to 6 DM-RS symbols per hop, grids of 7 / 10 / 13 symbols, any grid
    width from 1 to 275 PRB, timing advances, delays beyond the examined window, CFOs of kHz, other beta values."""
    grid = int(rng.choice([g for g in ((1, 2, 3, 5, 7, 13, 24, 52, 100, 270, 275) if wide else (6, 25, 52, 106, 273)) if g <= max_grid]))
    layers = int(rng.choice([1, 1, 1, 2, 3, 4]))
    masks = [SINGLE[rng.integers(len(SINGLE))]] if layers <= 2 else PAIRS[rng.integers(len(PAIRS))]
    n_hops = int(rng.choice([1, 1, 2]))
    n_prbs = int(rng.integers(1, grid + 1)) if rng.random() < 0.3 else int(rng.integers(1, min(grid, 12) + 1))
    interp = "cnn" if rng.random() < (0.2 if grid <= 52 else 0.08) else "linear"
    style = rng.choice(["split", "full", "partial"]) if n_hops == 2 else "split"
    scattered = rng.random() < 0.15
    hops = []
    for h in range(n_hops):
        lo, hi = (0, 14) if n_hops == 1 else ((0, 7) if h == 0 else (7, 14))
        nd = int(rng.integers(1, 5)) if n_hops == 1 else int(rng.integers(1, 4))
        if wide and rng.random() < 0.3:
            nd = int(rng.integers(4, 7))
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
        mp = sorted(rng.choice(grid, size=n_prbs, replace=False).tolist()) if scattered else None
        hops.append(S.hop_spec(dm, int(rng.integers(0, grid - n_prbs + 1)), n_prbs, start, n_alloc, masks, mask_prbs=mp))
    if n_hops == 2 and rng.random() < 0.5:          # same DM-RS count in both hops (register path)
        k = min(len(hops[0]["dmrs_symbols"]), len(hops[1]["dmrs_symbols"]))
        hops[0]["dmrs_symbols"], hops[1]["dmrs_symbols"] = hops[0]["dmrs_symbols"][:k], hops[1]["dmrs_symbols"][:k]
    n_sym = 14 if rng.random() < 0.88 else 12       # 12: element-wise writer; no CFO ramp possible (T:928-929)
    if wide and rng.random() < 0.25:
        n_sym = int(rng.choice([7, 10, 13]))
    if n_sym != 14:
        for h in hops:
            h["dmrs_symbols"] = sorted({min(s, n_sym - 1) for s in h["dmrs_symbols"]})
            h["start_symbol"] = min(h["start_symbol"], n_sym - 1)
            h["n_alloc"] = min(h["n_alloc"], n_sym - h["start_symbol"])
        if n_hops == 2 and set(hops[0]["dmrs_symbols"]) & set(hops[1]["dmrs_symbols"]):
            n_sym = 14
            for h in hops:                           # restore a valid 14-symbol description
                h["n_alloc"] = min(h["n_alloc"], 14 - h["start_symbol"])
    smoothing = str(rng.choice(["none", "mean", "filter", "filter", "mmse"])) if interp == "linear" else str(rng.choice(["none", "mean", "filter"]))
    any_cfo = any(len(h["dmrs_symbols"]) >= 2 for h in hops)
    cfo_comp = bool(rng.random() < 0.8) and not (n_sym != 14 and any_cfo)   # the reference cannot ramp a grid of other than 14 symbols
    case = S.case_spec("fuzz", grid, hops, n_layers=layers, smoothing=smoothing, n_sym=n_sym, cfo_compensate=cfo_comp,
                       scs=float(rng.choice([15e3, 30e3, 60e3])), seed=int(rng.integers(1 << 30)),
                       cfo_hz=float(rng.uniform(-400, 400)), delay_ns=float(rng.uniform(0, 400)))
    if wide:
        case["delay_ns"] = float(rng.choice([rng.uniform(-400, 400), rng.uniform(-400, 0), rng.uniform(1500, 6000)]))
        case["cfo_hz"] = float(rng.choice([rng.uniform(-400, 400), rng.uniform(-4000, 4000)]))
        case["beta"] = float(rng.choice([1.4125, 0.5, 1.0, 2.0]))
    extras = dict(interp=interp, layout_ref=bool(rng.random() < 0.3),
                  cnn_alpha=float(rng.choice([0.0, 0.4])) if interp == "cnn" else None,
                  mmse=(float(rng.choice([0.3e-6, 1.2e-6])), float(rng.choice([0.01, 0.1]))) if smoothing == "mmse" else None)
    return case, extras


def realize(case, extras, n_items=2):
    b = S.build_case(case, n_items)
    if extras.get("cnn_alpha") is not None:
        b.config.CNNSmoothingAlpha = extras["cnn_alpha"]
    if extras.get("mmse") is not None:              # extension: checked against its own oracle
        b.config.MMSEDelaySpread, b.config.MMSENoiseToSignal = extras["mmse"]
    return b


def compare_item(case, b, got_ch, got_sc, ref, stages, what, ta_ref=None):
    """The suite's protocol (conftest.check_outputs) plus what random narrow bands need: a TA neighbour bin the ORACLE's
    own transform puts within TA_TIE_RATIO of its arg-max is accepted (one bin per hop); 1-2 pilots give a flat /
    periodic |IFFT| whose arg-max is arbitrary on every side; the CFO has a float32 floor of ~3e-7 rad whatever the
    angle; "mean" smoothing can cancel to a small band mean, so its rounding scales with |H| ~ 1, not with the result.
    `ta_ref` = `(seconds, bins[hop], powers[hop][5])` from the REAL reference (conftest.fuzz_ta_reference): the TA
    expectation and the tie alternatives then come from the reference's own transform, the oracle's are not consulted."""
    n_pil, n_hops, scs = b.pilots.shape[0], len(case["hops"]), case["scs"]
    rs = [ref[1], ref[2], ref[3], ref[4], np.nan if ref[5] is None else ref[5]]
    got = list(got_sc)
    for j in range(5):                               # the same KIND of non-finite value (1 pilot: noise = residual / 0 = +inf, or 0 / 0 = NaN) counts as equal
        if not np.isfinite(rs[j]) and not np.isfinite(got[j]):
            same = (np.isnan(rs[j]) and np.isnan(got[j])) or (np.isinf(rs[j]) and np.isinf(got[j]) and np.sign(rs[j]) == np.sign(got[j]))
            # one pilot in all (noise denominator 0, T:913-915): residual / 0 is +inf or NaN by whether the residual's rounding left exactly 0
            same = same or (j == 0 and n_pil * sum(len(h["dmrs_symbols"]) for h in case["hops"]) * ((case["n_layers"] + 1) // 2) == 1
                            and not np.isneginf(rs[j]) and not np.isneginf(got[j]))
            assert same, f"{what}: scalar {j}: {got[j]} vs {rs[j]}"
            rs[j] = got[j] = 0.0 if j != 4 else np.nan
    alts = []
    if n_pil <= 2:
        # parity unpinned for this class: a flat / periodic |IFFT| has no defined arg-max.  What still must
        # hold: every hop's bin is one of the 288 examined ones, i.e. TA * 4096 * scs (* 2 with two hops) is an integer sum of
        # bins in [-144, 143] per hop
        tb = float(got[3]) * 4096.0 * scs * (2.0 if n_hops == 2 else 1.0)
        assert abs(tb - round(tb)) < 1e-6 and -144 * n_hops <= round(tb) <= 143 * n_hops, f"{what}: TA {got[3]!r} is not a sum of examined bins"
        got[3] = rs[3]
    elif ta_ref is not None:
        rs[3] = ta_ref[0]
        alts = ta_alternatives_from(ta_ref[1][:n_hops], ta_ref[2][:n_hops], scs)
    else:
        alts = ta_alternatives_from([st["ta_bin"] for st in stages], [st["ta_pw"] for st in stages], scs)
    if np.isfinite(rs[4]) and abs(got[4] - rs[4]) <= 5e-8 * scs:
        got[4] = rs[4]
    cond = max(1.0, 0.7 / float(np.abs(ref[0]).max())) if case["smoothing"] == "mean" else 1.0
    # ... and so does the RSRP, which for "mean" is |band mean|^2: twice the mean's relative rounding
    check_outputs(got_ch, got, ref[0], rs, 2e-5 * cond, 2e-5, what, alts, tol_rsrp=2e-5 * (2.0 * cond if cond > 1.0 else 1.0))
    return bool(alts) and got[3] != rs[3]
