"""CPU-side checks: the C-ABI library loads and exports every symbol include/ce_hip.h declares
(no compute calls without a GPU), and the host glue raises the reference's exception types
before anything touches a device."""
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, load_fixture

from srsran_ce_pytorch_amd import _lib, estimator as E


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = (ROOT / "include" / "ce_hip.h").read_text()
    declared = set(re.findall(r"^\s*(?:const\s+char\s*\*|int|void)\s+(ce_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.ce_abi_version() == _lib.CE_ABI_VERSION


def test_struct_layout_matches_header_sizes(lib):
    # sizes implied by include/ce_hip.h on LP64: hop 14+2*2 (+pad) +8+8(ptr)+8 = 48; plan 40+16+112+8+16+96 = 288
    import ctypes as C
    assert C.sizeof(_lib.HopDesc) == 48
    assert C.sizeof(_lib.PlanDesc) == 288
    assert C.sizeof(_lib.PlanInfo) == 40


def test_descriptor_errors_without_gpu(lib):
    import ctypes as C
    d = _lib.PlanDesc()
    h = C.c_void_p()
    assert lib.ce_plan_create(C.byref(d), C.byref(h)) == _lib.CE_ERR_INVALID     # abi_version 0
    assert b"ABI" in lib.ce_last_error()
    d.abi_version = _lib.CE_ABI_VERSION
    d.n_layers, d.n_hops, d.n_prb_grid, d.n_sym = 9, 1, 52, 14
    assert lib.ce_plan_create(C.byref(d), C.byref(h)) == _lib.CE_ERR_UNSUPPORTED
    d.n_layers, d.n_prb_grid = 1, 400
    assert lib.ce_plan_create(C.byref(d), C.byref(h)) == _lib.CE_ERR_UNSUPPORTED  # 4800 sc > 4096-point IFFT
    d.n_prb_grid, d.smoothing = 52, 7
    assert lib.ce_plan_create(C.byref(d), C.byref(h)) == _lib.CE_ERR_INVALID
    assert b"Unknown smoothing" in lib.ce_last_error()


def test_reference_error_conventions(lib):
    fx = load_fixture("hop2_2dmrs_each")
    fx.config.Smoothing = "bogus"
    with pytest.raises(ValueError, match="Unknown smoothing strategy bogus"):        # T:668
        E.make_plan(fx.hop1, fx.hop2, fx.config, fx.beta, 1, 52, 14)
    fx.config.Smoothing = "filter"
    fx.config.CyclicPrefixDurations = fx.config.CyclicPrefixDurations[:10]
    with pytest.raises(ValueError, match="length >= 14"):                            # T:816
        E.make_plan(fx.hop1, fx.hop2, fx.config, fx.beta, 1, 52, 14)
    fx = load_fixture("hop2_2dmrs_each")
    fx.hop2.DMRSsymbols = fx.hop1.DMRSsymbols.copy()
    with pytest.raises(AssertionError, match="Hops should not overlap"):             # T:862
        E.make_plan(fx.hop1, fx.hop2, fx.config, fx.beta, 1, 52, 14)
    fx = load_fixture("hop2_2dmrs_each")
    fx.hop2.DMRSREmask = ~fx.hop1.DMRSREmask
    with pytest.raises(AssertionError, match="DM-RS mask should be the same"):       # T:869
        E.make_plan(fx.hop1, fx.hop2, fx.config, fx.beta, 1, 52, 14)


def test_no_product_import_of_oracle():
    """The product package must never import, call or link anything under oracle/."""
    for f in (ROOT / "srsran_ce_pytorch_amd").rglob("*"):
        if f.suffix in {".py", ".hip", ".h", ".cpp"}:
            txt = f.read_text()
            assert "ce_oracle" not in txt and "oracle/" not in txt.replace("the oracle/", ""), f


# ------------------------------------------------------------------------------------------------------------
# Host-side derivations of the C++ library (ce_plan_derive_host: no GPU) against the oracle / brute force
# ------------------------------------------------------------------------------------------------------------
import ce_oracle as O  # noqa: E402
from conftest import golden_names  # noqa: E402
from srsran_ce_pytorch_amd import synth as S  # noqa: E402


@pytest.mark.parametrize("name", golden_names("T"))
def test_host_derivations_match_oracle(lib, name):
    fx = load_fixture(name)
    L = fx.pilots.shape[2]
    v = E.derive_host(fx.hop1, fx.hop2, fx.config, fx.beta, L, fx.case["n_prb_grid"], fx.case["n_sym"])
    hops = [fx.hop1] + ([fx.hop2] if len(fx.case["hops"]) > 1 else [])
    assert v.n_re == fx.pilots.shape[0] and v.n_dmrs_total == fx.pilots.shape[1]
    # symbol start times (T:809-820) and CFO sample span (T:418-426)
    np.testing.assert_allclose(np.array(v.sst[:]), O.symbol_start_time(fx.config.CyclicPrefixDurations, fx.config.scs), rtol=1e-15)
    for h, hop in enumerate(hops):
        ix = np.flatnonzero(hop.DMRSsymbols)
        if ix.size >= 2:
            cpd = fx.config.CyclicPrefixDurations * (fx.config.scs / 1000.0)
            n_samples = float(ix[1] - ix[0]) + float(np.sum(cpd[ix[0] + 1: ix[1] + 1]))
            assert v.two_pi_nsamples[h] == pytest.approx(2 * np.pi * n_samples, rel=1e-15)
    assert bool(v.cfo_estimated) == any(np.flatnonzero(h.DMRSsymbols).size >= 2 for h in hops)
    # RC taps and virtual-pilot count (T:638-647, T:184-234)
    if fx.config.Smoothing == "filter":
        dpp = int(fx.hop1.DMRSREmask[:, 0].sum())
        n_act = int(fx.hop1.maskPRBs.sum())
        rc = O.get_rc_filter(12 // dpp, min(3, n_act))
        assert v.rc_len == rc.size
        np.testing.assert_allclose(np.array(v.rc[: rc.size]), rc, rtol=0, atol=2e-16)
        assert v.n_pils == (min(12, rc.size // 2) if n_act > 1 else dpp)
    # normalisers (T:901-915)
    n_pilots = fx.hop1.nPRBs * int(fx.hop1.DMRSREmask[:, 0].sum()) * fx.pilots.shape[1]
    assert v.n_pilots == n_pilots and v.noise_den == ((L + 1) // 2) * n_pilots - 1
    # interpolation anchors / weights (T:311-338) against a brute-force searchsorted over the hop band
    for h, hop in enumerate(hops):
        for c in range((L + 1) // 2):
            mask_all = np.tile(hop.DMRSREmask[:, c], hop.nPRBs)
            filled = np.flatnonzero(mask_all)
            dpp = int(hop.DMRSREmask[:, c].sum())
            assert v.last_idx[h][c] == filled[-1]
            for p in range(filled[0] + 1, filled[-1]):
                ro = int(np.searchsorted(filled, p, side="left"))
                q, r = divmod(p, 12)
                assert q * dpp + v.r_ord[h][c][r] == ro
                lp, rp = filled[ro - 1], filled[ro]
                assert v.alpha[h][c][r] == np.float32(p - lp) / np.float32(rp - lp)


def test_host_derivation_register_path_and_limits(lib):
    h1, h2, cfg = S.numpy_hops(S.bench_case("filter"))
    v = E.derive_host(h1, h2, cfg, 1.4125, 1, 273, 14)
    assert v.reg_nd == 2 and v.filt_windowed == 1 and v.ta_nres[0] == 8 and v.contig[0] == 1
    assert v.lds_bytes <= 53 * 1024            # three workgroups per CU
    h1, h2, cfg = S.numpy_hops(S.case_spec("x", 52, [S.hop_spec([2, 7, 11], 4, 5, re_masks=[S.TYPE2_CDM0, S.TYPE2_CDM1])], n_layers=3))
    v = E.derive_host(h1, h2, cfg, 1.0, 3, 52, 14)
    assert v.reg_nd == 0 and v.ta_nres[0] == 16 and v.filt_windowed == 0
    with pytest.raises(NotImplementedError):
        h1, h2, cfg = S.numpy_hops(S.case_spec("too_wide", 400, [S.hop_spec([2, 11], 0, 400)]))
        E.derive_host(h1, h2, cfg, 1.0, 1, 400, 14)
    # non-contiguous maskPRBs (allowed by the reference: extraction uses maskPRBs, fill uses PRBstart/nPRBs)
    h1, h2, cfg = S.numpy_hops(S.case_spec("nc", 52, [S.hop_spec([2, 11], 4, 6)]))
    h1.maskPRBs = np.zeros(52, bool); h1.maskPRBs[[4, 5, 6, 20, 21, 22]] = True
    assert E.derive_host(h1, h2, cfg, 1.0, 1, 52, 14).contig[0] == 0


def test_wave_per_item_kernel_takes_at_most_four_dmrs_symbols_per_hop(lib):
    """The reference accepts any DMRSsymbols mask (T:564-568); the wave-per-item kernel's per-hop phasor tables hold four
    symbols, so a narrow hop with five of them must keep the workgroup kernels (tests/test_hip_parity.py runs both)."""
    for dm, narrow in (([2, 5, 8, 11], 1), ([1, 3, 6, 9, 12], 0)):
        h1, h2, cfg = S.numpy_hops(S.case_spec("x", 52, [S.hop_spec(dm, 40, 3)]))
        assert E.derive_host(h1, h2, cfg, 1.0, 1, 52, 14).narrow == narrow


def test_shipped_library_ignores_tuning_knobs(lib, monkeypatch):
    """include/ce_hip.h "Tuning knobs": only the diagnostic build (libce_hip_knobs.so, -DCE_TUNING_KNOBS) reads the
    environment at plan creation; a stray variable in a user's environment cannot change the shipped library's kernel
    selection or LDS sizing."""
    import ctypes as C
    h1, h2, cfg = S.numpy_hops(S.bench_case("filter"))
    desc, _, keep = E.build_desc(h1, h2, cfg, 1.4125, 1, 273, 14)
    knobs = C.CDLL(str(_lib.KNOBS_LIB_PATH))
    knobs.ce_plan_derive_host.argtypes = [C.POINTER(_lib.PlanDesc), C.POINTER(_lib.PlanHostView)]
    v0, v1, v2 = _lib.PlanHostView(), _lib.PlanHostView(), _lib.PlanHostView()
    assert knobs.ce_plan_derive_host(C.byref(desc), C.byref(v0)) == 0 and v0.reg_nd == 2
    monkeypatch.setenv("CE_FORCE_GENERIC", "1")
    monkeypatch.setenv("CE_LDS_PAD_BYTES", "4096")
    assert lib.ce_plan_derive_host(C.byref(desc), C.byref(v1)) == 0
    assert v1.reg_nd == 2 and v1.lds_bytes == v0.lds_bytes                      # shipped: environment ignored
    assert knobs.ce_plan_derive_host(C.byref(desc), C.byref(v2)) == 0
    assert v2.reg_nd == 0 and v2.lds_bytes >= v0.lds_bytes + 4096 - 16          # diagnostic build: knobs honoured
    src = (ROOT / "srsran_ce_pytorch_amd" / "csrc" / "ce_api.hip").read_text()
    assert src.count("getenv(") == 1                                            # the one call inside ce_knob(), under #ifdef CE_TUNING_KNOBS
    header = (ROOT / "include" / "ce_hip.h").read_text()
    for name in set(re.findall(r'ce_knob\("(CE_\w+)"\)', src)):
        assert name in header, f"{name} not documented in include/ce_hip.h"
    del keep


def test_mmse_extension_host_filter_matches_oracle(lib):
    """EXTENSION (parity unpinned: no reference counterpart): the C++ LU solve of W = R (R + nsr I)^-1 against numpy."""
    for case, L in ((S.bench_case("mmse"), 1), (S.case_spec("m2", 52, [S.hop_spec([2, 11], 4, 3)], smoothing="mmse"), 1),
                    (S.case_spec("m3", 52, [S.hop_spec([2, 11], 4, 20, re_masks=[S.TYPE2_CDM0])], smoothing="mmse"), 1)):
        h1, h2, cfg = S.numpy_hops(case)
        cfg.MMSEDelaySpread, cfg.MMSENoiseToSignal = 1.5e-6, 0.02
        v = E.derive_host(h1, h2, cfg, 1.0, L, case["n_prb_grid"], 14)
        sc = np.flatnonzero(np.kron(h1.maskPRBs, h1.DMRSREmask[:, 0]))
        m = min(32, sc.size)
        w = O.mmse_matrix(sc[:m], cfg.scs, 1.5e-6, 0.02)
        got = np.array(v.mmse_w[0])[:m, :m] + 1j * np.array(v.mmse_w[1])[:m, :m]
        assert np.abs(got - w).max() < 2e-6 * np.abs(w).max()
        assert not np.array(v.mmse_w[0])[m:].any() and not np.array(v.mmse_w[0])[:, m:].any()
    h1, h2, cfg = S.numpy_hops(S.case_spec("m4", 52, [S.hop_spec([2, 11], 4, 8, re_masks=[[1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]])], smoothing="mmse"))
    with pytest.raises(NotImplementedError):      # 5 bunched pilots per PRB, 40 pilots: the anchored last block starts mid-PRB
        E.derive_host(h1, h2, cfg, 1.0, 1, 52, 14)


def test_compat_alias_modules_expose_the_reference_names():
    """`from ce_rule_tensorized import EstimatorConfig, HopConfig, srs_channel_estimator` (validate_all.py:18) and the
    same from `srs_estimator_torch` (validate_case0.py:12) must resolve to this build with compat/ on the path."""
    import importlib
    import sys
    sys.path.insert(0, str(ROOT / "compat"))
    try:
        for mod in ("ce_rule_tensorized", "srs_estimator_torch", "ce_rule_baseline", "ce_dl_cnn"):
            sys.modules.pop(mod, None)
            m = importlib.import_module(mod)
            if mod == "ce_dl_cnn":      # the reference's third variant: same signature, in-painting interpolation
                assert m.srs_channel_estimator.func is E.srs_channel_estimator and m.srs_channel_estimator.keywords == {"interp": "cnn"}
            else:
                assert m.srs_channel_estimator is E.srs_channel_estimator
            hop = m.HopConfig(DMRSsymbols=[1] + [0] * 13, DMRSREmask=[[1]] * 12, PRBstart=0, nPRBs=1, maskPRBs=[1], startSymbol=0, nAllocatedSymbols=14)
            cfg = m.EstimatorConfig(scs=15e3, CyclicPrefixDurations=[0.0] * 14)
            assert cfg.Smoothing == "filter" and cfg.CFOCompensate is True and hop.nPRBs == 1      # defaults of T:24-29
    finally:
        sys.path.remove(str(ROOT / "compat"))
        for mod in ("ce_rule_tensorized", "srs_estimator_torch", "ce_rule_baseline", "ce_dl_cnn"):
            sys.modules.pop(mod, None)
