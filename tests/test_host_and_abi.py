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
    # sizes implied by include/ce_hip.h on LP64: hop 14+2*2 (+pad) +8+8(ptr)+8 = 48; plan 40+16+112+8+96 = 272
    import ctypes as C
    assert C.sizeof(_lib.HopDesc) == 48
    assert C.sizeof(_lib.PlanDesc) == 272
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
