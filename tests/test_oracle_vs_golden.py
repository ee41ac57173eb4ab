"""Pin the CPU oracle to the real reference: every committed fixture (inputs + outputs of
``ce_rule_tensorized`` / ``ce_dl_cnn`` run in the build container, tools/make_golden.py)."""
import numpy as np
import pytest

from conftest import check_outputs, golden_names, load_fixture, ta_tie_alternatives

import ce_oracle as O
import ce_oracle_baseline as OB

TOL_CH = 2e-6    # both sides are complex64 pipelines; measured <= 5e-7
TOL_SC = 2e-6


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(name):
    fx = load_fixture(name)
    interp = "cnn" if fx.variant == "C" else "linear"
    for it in range(fx.grids.shape[0]):
        out = O.srs_channel_estimator(fx.grids[it], fx.pilots, fx.beta, fx.hop1, fx.hop2, fx.config, interp=interp)
        sc = [out[1], out[2], out[3], out[4], np.nan if out[5] is None else out[5]]
        alts = ta_tie_alternatives(fx, it) if fx.variant == "N" else ()     # near-tie class: see conftest.TA_TIE_RATIO
        check_outputs(out[0], sc, fx.ref_ch_est[it], fx.ref_scalars[it], TOL_CH, TOL_SC, f"{name}[{it}]", alts)


@pytest.mark.parametrize("name", golden_names("T"))
def test_loop_baseline_matches_reference_fixture(name):
    """oracle/ce_oracle_baseline.py (the loop-style restatement of src/ce_rule_baseline.py that bench.py times as the
    CPU baseline) against the same fixtures: the reference's own baseline and tensorized forms agree to <= 4.2e-8 on
    them (tests/golden/MANIFEST.json: baseline_vs_tensorized_max_abs)."""
    fx = load_fixture(name)
    for it in range(fx.grids.shape[0]):
        out = OB.srs_channel_estimator(fx.grids[it], fx.pilots, fx.beta, fx.hop1, fx.hop2, fx.config)
        sc = [out[1], out[2], out[3], out[4], np.nan if out[5] is None else out[5]]
        check_outputs(out[0], sc, fx.ref_ch_est[it], fx.ref_scalars[it], TOL_CH, TOL_SC, f"{name}[{it}]/loop")


def test_rc_filter_known_taps():
    """Tap values quoted in SURVEY.md section 8a for stride 2 / 3 RB (15 taps)."""
    rc = O.get_rc_filter(2, 3)
    assert rc.size == 15 and abs(rc.sum() - 1) < 1e-15
    np.testing.assert_allclose(rc[:8], [-0.039159, -0.028799, 0, 0.044525, 0.097072, 0.146705, 0.182153, 0.195006], atol=1e-6)
    assert [O.get_rc_filter(*a).size for a in [(2, 1), (2, 2), (1, 3), (3, 3)]] == [5, 11, 31, 11]


def test_error_conventions():
    with pytest.raises(ValueError):
        O.get_rc_filter(0, 3)
    with pytest.raises(ValueError):
        O.create_virtual_pilots(np.ones(3, np.complex64), -1)
    fx = load_fixture("prb1_filter")
    fx.config.Smoothing = "bogus"
    with pytest.raises(ValueError):
        O.srs_channel_estimator(fx.grids[0], fx.pilots, fx.beta, fx.hop1, fx.hop2, fx.config)


@pytest.mark.parametrize("name", golden_names("M"))
def test_complex64_estimate_of_a_complex128_grid_against_the_reference(name):
    """The reference's outputs for complex128 grids (pilots complex64 as its harness passes them, or complex128) against a
    complex64 estimate of the same slot -- the oracle's here, the HIP path's in tests/test_hip_parity.py: the pinned
    distance of the deliberate narrowing."""
    fx = load_fixture(name)
    for it in range(fx.grids.shape[0]):
        out = O.srs_channel_estimator(fx.grids[it], fx.pilots, fx.beta, fx.hop1, fx.hop2, fx.config)
        got = [out[1], out[2], out[3], out[4], np.nan if out[5] is None else out[5]]
        check_outputs(out[0], got, fx.ref_ch_est[it], fx.ref_scalars[it], 1e-6, 5e-6, f"{name}[{it}]")


def test_oracle_time_alignment_against_the_reference_on_the_fuzz_slice():
    """tests/golden/fuzz_ta_reference.npz holds the REAL reference's arg-max bins and IFFT powers for the GPU suite's fuzz
    cases (tools/make_fuzz_ta_reference.py).  A sample of them on the CPU: the numpy oracle, put through the same protocol the
    HIP path is held to, must match the reference's TA or sit on a neighbour the reference's own transform puts within
    TA_TIE_RATIO of its maximum."""
    import fuzz_cases as F
    import test_hip_fuzz as TF
    from conftest import fuzz_ta_reference

    checked = 0
    for idx in range(0, TF.N_CASES, 9):
        case, extras = F.draw(np.random.default_rng([TF.BASE_SEED, idx]), 273 if idx % 8 == 0 else 106)
        tr = fuzz_ta_reference(idx)
        assert (tr is None) == (case["smoothing"] == "mmse")
        if tr is None:
            continue
        b = F.realize(case, extras)
        for it in range(2):
            st = []
            ref = O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config, interp=extras["interp"], stages=st)
            got = [ref[1], ref[2], ref[3], ref[4], np.nan if ref[5] is None else ref[5]]
            F.compare_item(case, b, ref[0], got, ref, st, f"fuzz[{idx}][{it}]", tr[it])
            checked += 1
    assert checked >= 60
