"""Shared test plumbing: the ``gpu`` marker, fixture loading, path setup.

``tests/`` is one of the three places allowed to import ``oracle/`` (the checker)."""
from __future__ import annotations

import json
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN_DIR = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(variant=None):
    names = []
    for f in sorted(GOLDEN_DIR.glob("*.npz")):
        with np.load(f) as z:
            v = str(z["variant"])
        # "F": fuzz_ta_reference.npz, a table of reference bins, not a slot fixture; "M": complex128-grid fixtures, only on request
        if v == "F" or (v == "M" and variant != "M") or (variant is not None and v != variant):
            continue
        names.append(f.stem)
    return names


def load_fixture(name: str) -> SimpleNamespace:
    """Rebuild the exact input the reference saw (zeros outside the stored DM-RS columns)."""
    from srsran_ce_pytorch_amd import synth as S

    with np.load(GOLDEN_DIR / f"{name}.npz") as z:
        case = json.loads(str(z["case_json"]))
        cols = z["cols"]
        gc = z["grid_cols"]
        grids = np.zeros((gc.shape[0], gc.shape[1], case["n_sym"]), np.complex64)
        grids[:, :, cols] = gc
        fx = SimpleNamespace(case=case, variant=str(z["variant"]), pilots=z["pilots"], grids=grids,
                             ref_ch_est=z["ref_ch_est"], ref_scalars=z["ref_scalars"])
        if "ta_bin" in z:                                   # variant "N": the reference's own arg-max bins and the power around them
            fx.ta_bin, fx.ta_pw = z["ta_bin"], z["ta_pw"]
        if "pilots_complex128" in z:                        # variant "M": the reference was given a complex128 grid (and these pilots' dtype)
            fx.pilots_complex128 = bool(z["pilots_complex128"])
    hops = [S._hop_arrays(case, h) for h in case["hops"]]
    fx.hop1 = hops[0]
    fx.hop2 = hops[1] if len(hops) > 1 else S.empty_hop_arrays()
    fx.config = SimpleNamespace(scs=case["scs"], CyclicPrefixDurations=S.normal_cp_ms(case["scs"]),
                                Smoothing=case["smoothing"], CFOCompensate=case["cfo_compensate"])
    if "cnn_alpha" in case:
        fx.config.CNNSmoothingAlpha = case["cnn_alpha"]
    fx.beta = case["beta"]
    return fx


# Near-tie class of the time alignment (fixtures of variant "N", tools/make_ta_neartie.py): where the reference's own IFFT
# puts a neighbouring bin within this relative power of its arg-max, float32 rounding of the transform decides the bin
# (the numpy oracle itself disagrees with torch.fft.ifft on 3 of the 42 committed items), so that neighbour is accepted too.
TA_TIE_RATIO = 1e-5


def ta_tie_alternatives(fx, item):
    """TA values that differ from the reference's by ONE bin per hop towards a neighbour whose power, in the
    reference's own transform, is within TA_TIE_RATIO of the winner's -- computed with the reference's arithmetic
    (T:698, T:918-919) so the comparison stays exact.  Neighbours are taken among the 288 examined bins; bin 4095 (the
    advance side's last) and bin 0 (the delay side's first) are neighbours as well: a peak between them is decided by
    the reference's `>=` between the two sides' maxima (T:693)."""
    scs, bins, n_hops = float(fx.case["scs"]), [int(b) for b in fx.ta_bin[item]], len(fx.case["hops"])
    return ta_alternatives_from(bins, [[float(x) for x in fx.ta_pw[item][h]] for h in range(n_hops)], scs)


def ta_alternatives_from(bins, powers, scs):
    """All TA values obtained by moving, in each hop independently, the chosen bin to a neighbour whose power (in the
    transform the powers come from; `powers[h]` = the chosen bin's power in the MIDDLE, its neighbours at distance 1 -- and
    2, when five values are given -- on either side, in the window of the 288 examined bins, where bins 4095 and 0 are
    adjacent; negative = outside the window) is within TA_TIE_RATIO of the chosen bin's -- every combination except
    "nothing moved" (two one-PRB hops can each sit on a tie: the long-form fuzzer met one)."""
    import itertools
    n_hops = len(bins)
    moves = []
    for h in range(n_hops):
        pw = [float(x) for x in powers[h]]
        mid = len(pw) // 2
        moves.append([0] + [d for d in range(-mid, mid + 1) if d and pw[mid + d] >= (1.0 - TA_TIE_RATIO) * pw[mid] and pw[mid + d] >= 0.0])
    alts = []
    for combo in itertools.product(*moves):
        if not any(combo):
            continue
        ta = 0.0
        for k in range(n_hops):
            ta = ta + float(bins[k] + combo[k]) / 4096.0 / float(scs)
        alts.append(ta / 2.0 if n_hops == 2 else ta)
    return alts


_FUZZ_TA = None


def fuzz_ta_reference(idx):
    """The REAL reference's time alignment for case `idx` of the GPU suite's fuzz slice (tests/golden/fuzz_ta_reference.npz,
    tools/make_fuzz_ta_reference.py): `None` where the reference has no answer (the "mmse" extension), else per item
    `(ta_seconds, bins[hop], powers[hop][5])` -- the reference's own arg-max bins and its own IFFT powers around them."""
    global _FUZZ_TA
    if _FUZZ_TA is None:
        with np.load(GOLDEN_DIR / "fuzz_ta_reference.npz") as z:
            _FUZZ_TA = {k: z[k] for k in ("ta_bin", "ta_pw", "ta_ref", "valid")}
    z = _FUZZ_TA
    if not z["valid"][idx]:
        return None
    return [(float(z["ta_ref"][idx, it]), [int(x) for x in z["ta_bin"][idx, it]], z["ta_pw"][idx, it].astype(np.float64)) for it in range(2)]


def check_outputs(got_ch, got_scalars, ref_ch, ref_scalars, tol_ch, tol_sc, what="", ta_alternatives=(), tol_rsrp=None):
    """Comparison protocol used everywhere: channel estimate error relative to the largest
    reference magnitude; scalars relative, except the residual noise whose floor is rounding
    noise of the EPRE (it is a difference of nearly equal quantities when nothing is smoothed);
    TA must be bit-identical (an integer bin index through the reference's two float64 divisions, T:698);
    cfo NaN <=> "not estimated"."""
    scale = float(np.abs(ref_ch).max())
    if scale == 0.0:                                     # an all-zero grid (fixtures zero_grid*): the estimate must be exactly zero too
        assert not np.any(got_ch), f"{what}: the reference's estimate is all zeros, this one is not"
        scale = 1.0
    err = float(np.abs(got_ch - ref_ch).max()) / scale
    noise, rsrp, epre, ta, cfo = [float(x) for x in got_scalars]
    r_noise, r_rsrp, r_epre, r_ta, r_cfo = [float(x) for x in ref_scalars]
    # An estimate that is the small remainder of cancelling terms -- the reference's own RSRP under 1/16 of its EPRE: a CFO of kHz
    # aliases the hop's CFO estimate and the de-rotation turns the DM-RS symbols against each other (fixture cfo_alias_cancel_7prb:
    # EPRE 8.3, RSRP 5e-4; the numpy oracle and the REAL reference differ by 3.0e-5 of max|h| there) -- carries the rounding of
    # the amplitude that went in, sqrt(EPRE), not of the little that came out, sqrt(RSRP): the tolerance scales by their ratio.
    cond = 1.0
    if np.isfinite(r_rsrp) and np.isfinite(r_epre) and 0.0 < 16.0 * r_rsrp < r_epre:
        cond = min(1000.0, float(np.sqrt(r_epre / r_rsrp)))
    assert err <= tol_ch * cond, f"{what}: ch_est rel-max err {err:.3e} > {tol_ch * cond:.1e}"
    assert abs(rsrp - r_rsrp) <= (tol_sc if tol_rsrp is None else tol_rsrp) * (2.0 * cond if cond > 1.0 else 1.0) * abs(r_rsrp), f"{what}: rsrp {rsrp} vs {r_rsrp}"
    assert abs(epre - r_epre) <= tol_sc * abs(r_epre), f"{what}: epre {epre} vs {r_epre}"
    assert abs(noise - r_noise) <= tol_sc * max(abs(r_noise), 1e-2 * abs(r_epre)), f"{what}: noise {noise} vs {r_noise}"
    assert ta == r_ta or ta in ta_alternatives, f"{what}: time alignment {ta!r} vs {r_ta!r}"      # index work: bit-exact (T:698)
    if np.isnan(r_cfo):
        assert np.isnan(cfo), f"{what}: cfo should be 'not estimated'"
    else:
        assert abs(cfo - r_cfo) <= tol_sc * max(abs(r_cfo), 1.0), f"{what}: cfo {cfo} vs {r_cfo}"
    return err
