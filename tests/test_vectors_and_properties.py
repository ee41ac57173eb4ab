"""CPU tests: srsRAN vector file formats (round trip), and property tests of the oracle itself (hypothesis):
the invariants the full-size GPU tests rely on must hold for the reference algorithm."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import ce_oracle as O
from srsran_ce_pytorch_amd import synth as S, vectors as V


def test_entry_file_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    grid = np.zeros((624, 14, 2), np.complex64)
    idx = rng.choice(624 * 14 * 2, 500, replace=False)
    grid.reshape(-1)[idx] = (rng.standard_normal(500) + 1j * rng.standard_normal(500)).astype(np.complex64)
    f = tmp_path / "rg.dat"
    V.write_entries(f, *V.grid_to_entries(grid))
    assert f.stat().st_size == 500 * 12
    ent = V.read_entries(f)
    assert np.array_equal(V.entries_to_grid(ent, 624, 14), grid)
    assert V.compare_at_entries(grid, ent) == (0.0, 0.0)
    (tmp_path / "bad.dat").write_bytes(b"\0" * 13)
    with pytest.raises(ValueError):
        V.read_entries(tmp_path / "bad.dat")


def test_pilot_file_orders(tmp_path):
    rng = np.random.default_rng(1)
    p = (rng.standard_normal((18, 4, 2)) + 1j * rng.standard_normal((18, 4, 2))).astype(np.complex64)   # [re, sym, layer]
    for order, perm in (("sym-re-layer", (1, 0, 2)), ("layer-sym-re", (2, 1, 0)), ("re-sym-layer", (0, 1, 2))):
        f = tmp_path / f"{order}.dat"
        np.ascontiguousarray(p.transpose(perm)).tofile(f)
        assert np.array_equal(V.read_pilots(f, 4, 18, 2, order), p)


def _case(n_prbs, smoothing, layers, seed):
    masks = [S.TYPE1_CDM0] if layers <= 2 else [S.TYPE1_CDM0, S.TYPE1_CDM1]
    return S.case_spec("prop", 52, [S.hop_spec([2, 11], 3, n_prbs, re_masks=masks)], n_layers=layers, smoothing=smoothing, seed=seed)


@settings(max_examples=12, deadline=None, derandomize=True)
@given(n_prbs=st.integers(3, 30), smoothing=st.sampled_from(["none", "mean", "filter"]), layers=st.integers(1, 4), seed=st.integers(0, 10_000))
def test_oracle_scaling_and_rotation_invariants(n_prbs, smoothing, layers, seed):
    b = S.build_case(_case(n_prbs, smoothing, layers, seed), 1)
    base = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)
    twice = O.srs_channel_estimator(b.grids[0] * np.float32(2), b.pilots, b.beta, b.hop1, b.hop2, b.config)
    assert np.array_equal(twice[0], base[0] * np.float32(2))                    # exact: power-of-two scaling
    for i in (1, 2, 3):
        assert twice[i] == pytest.approx(4 * base[i], rel=1e-12)
    assert twice[4] == base[4] and twice[5] == pytest.approx(base[5], rel=1e-9, abs=1e-9)
    rot = np.complex64(np.exp(0.7j))
    turned = O.srs_channel_estimator(b.grids[0] * rot, b.pilots, b.beta, b.hop1, b.hop2, b.config)
    # float32 rounding scales with the pilots that were averaged, not with the result (a band mean can cancel to ~0.01)
    assert np.abs(turned[0] - base[0] * rot).max() <= 5e-6 * max(np.abs(base[0]).max(), np.abs(b.grids[0]).max())
    assert turned[4] == base[4]
    # zeros outside the allocation, hop band fully populated
    assert not base[0][: 36].any() and not base[0][12 * (3 + n_prbs):].any()
    assert np.all(np.abs(base[0][36: 12 * (3 + n_prbs)]) > 0)


def test_oracle_time_alignment_tracks_delay():
    for delay_ns in (0.0, 500.0, 1000.0):     # bins = delay * scs * 4096; the weaker late taps pull the peak <= 3 bins
        b = S.build_case(dict(_case(25, "none", 1, 3), delay_ns=delay_ns, noise_var=1e-6), 1)
        ta = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)[4]
        assert 0 <= ta * 4096 * 30e3 - delay_ns * 1e-9 * 30e3 * 4096 <= 3
