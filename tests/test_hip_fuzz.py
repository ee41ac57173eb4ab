"""A bounded, deterministic slice of the differential fuzzer in the GPU suite: N_CASES seeded random geometries (see
fuzz_cases.py for what is drawn) through the HIP estimator and the CPU oracle, compared with the suite's protocol; where
the oracle raises, the HIP boundary must raise the same exception class.  No input class is skipped.

The time alignment -- index work, decided by the reference's own complex64 IFFT -- is NOT taken from the oracle: the expected
value and the near-tie alternatives of every case the reference can run come from tests/golden/fuzz_ta_reference.npz
(the real ce_rule_tensorized / ce_dl_cnn run on these very cases, tools/make_fuzz_ta_reference.py); only the "mmse"
extension, which the reference lacks, falls back to the oracle's transform."""
import numpy as np
import pytest
import torch

import ce_oracle as O
import fuzz_cases as F
from conftest import fuzz_ta_reference
from srsran_ce_pytorch_amd import estimator as E

pytestmark = pytest.mark.gpu
N_CASES = 400
BASE_SEED = 20261004


@pytest.mark.parametrize("idx", range(N_CASES))
def test_fuzz_case(idx):
    rng = np.random.default_rng([BASE_SEED, idx])
    case, extras = F.draw(rng, 273 if idx % 8 == 0 else 106)
    _run_case(case, extras, idx)


# Cases the long-form fuzzer (tools/fuzz_parity.py) flagged, kept as drawn.  [0]: one PRB, four adjacent pilots, two layers --
# the powers of bins 4095 and 0 are EQUAL in the oracle's float32 transform (the reference's `>=` takes the delay side);
# the checker now treats those two bins as neighbours (conftest.ta_tie_alternatives).
LOGGED_CASES = [
    {"case": {"name": "fuzz", "n_prb_grid": 52, "hops": [{"dmrs_symbols": [8], "prb_start": 27, "n_prbs": 1, "start_symbol": 0, "n_alloc": 12,
                                                          "re_masks": [[0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]]}],
              "n_layers": 2, "smoothing": "filter", "cfo_compensate": True, "scs": 15000.0, "beta": 1.4125, "n_sym": 14, "seed": 296368357,
              "cfo_hz": 20.00408490627808, "delay_ns": 51.825950456196864, "noise_var": 0.005},
     "extras": {"interp": "linear", "layout_ref": False, "cnn_alpha": None, "mmse": None}},
    # [1]: two one-PRB hops (four adjacent pilots each), each on a tie of its own -- hop 1's bins 60 and 61 EQUAL in the oracle's
    # transform, hop 2's bins 48 and 49 within 2.4e-7: the checker now allows one tied-neighbour move PER HOP
    {"case": {"name": "fuzz", "n_prb_grid": 273, "hops": [{"dmrs_symbols": [0, 4, 6], "prb_start": 144, "n_prbs": 1, "start_symbol": 0, "n_alloc": 10, "re_masks": [[0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]]}, {"dmrs_symbols": [7], "prb_start": 184, "n_prbs": 1, "start_symbol": 5, "n_alloc": 9, "re_masks": [[0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]]}], "n_layers": 1, "smoothing": "filter", "cfo_compensate": True, "scs": 30000.0, "beta": 1.4125, "n_sym": 14, "seed": 718188862, "cfo_hz": 345.8305489523933, "delay_ns": 335.197121631845, "noise_var": 0.005}, "extras": {"interp": "linear", "layout_ref": False, "cnn_alpha": None, "mmse": None}},
]


@pytest.mark.parametrize("k", range(len(LOGGED_CASES)))
def test_logged_fuzz_case(k):
    _run_case(LOGGED_CASES[k]["case"], LOGGED_CASES[k]["extras"], N_CASES + k)   # (their rows of the reference file follow the seeded ones)


def _run_case(case, extras, idx):
    interp = extras["interp"]
    try:
        b = F.realize(case, extras)
    except Exception as e:                                   # a draw the generator itself cannot lay out
        pytest.fail(f"generator: {e!r} :: {case}")
    want, stages, werr = [], [], None
    try:
        for it in range(2):
            st = []
            want.append(O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config, interp=interp, stages=st))
            stages.append(st)
    except (ValueError, AssertionError, IndexError) as e:    # IndexError: a DM-RS mask shorter than the 14 symbol start times (T:440-447)
        werr = e
    dev = torch.device("cuda:0")
    g = torch.as_tensor(b.grids, device=dev)[None]
    if not extras["layout_ref"]:
        g = g.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    try:
        out = E.estimate(g, torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config, interp=interp)
        torch.cuda.synchronize()
    except NotImplementedError as e:
        # outside the build's stated limits (include/ce_hip.h: CE_ERR_UNSUPPORTED) -- only the mmse EXTENSION has such inputs here
        assert case["smoothing"] == "mmse", f"unsupported on a reference input: {e} :: {case} {extras}"
        return
    except (ValueError, AssertionError) as e:
        assert werr is not None, f"HIP raised {e!r}, the oracle did not :: {case} {extras}"
        assert type(e) is type(werr) or (isinstance(werr, IndexError) and isinstance(e, ValueError)), f"{e!r} vs {werr!r}"
        return
    assert werr is None, f"the oracle raised {werr!r}, HIP did not :: {case} {extras}"
    ch = out[0][0].cpu().numpy()
    sc = [t[0].cpu().numpy() if t.numel() else None for t in out[1:]]
    ta_ref = fuzz_ta_reference(idx)
    assert (ta_ref is None) == (case["smoothing"] == "mmse"), "the reference pins every case but the mmse extension's"
    for it in range(2):
        got = [sc[0][it], sc[1][it], sc[2][it], sc[3][it], np.nan if sc[4] is None else sc[4][it]]
        F.compare_item(case, b, ch[it], got, want[it], stages[it], f"fuzz[{idx}][{it}] {case} {extras}", None if ta_ref is None else ta_ref[it])
