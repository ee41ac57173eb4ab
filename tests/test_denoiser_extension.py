"""Conv2d denoiser EXTENSION (include/ce_denoise.h) -- "parity unpinned": nothing in the reference corresponds to it,
so the HIP kernel is checked against the build's own numpy restatement (oracle/ce_denoise_oracle.py) only.
Tolerance: both sides round activations to fp16 at the same points; what differs is the accumulation order (and, rarely,
an fp16 rounding boundary), so the correction agrees to ~1e-3 of its own size; stated here as 2e-3 of max|h|."""
import re

import numpy as np
import pytest

import ce_denoise_oracle as DO
from conftest import ROOT
from srsran_ce_pytorch_amd import _lib
from srsran_ce_pytorch_amd.denoiser import SHAPES, random_weights


def test_library_exports_the_denoiser_symbols():
    _lib.build()
    lib = _lib.load()
    header = (ROOT / "include" / "ce_denoise.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|void)\s+(ce_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTS_DENOISE)
    for sym in declared:
        assert hasattr(lib, sym)


def test_oracle_identity_and_locality():
    w = random_weights(1)
    zero = {k: np.zeros_like(v) for k, v in w.items()}
    rng = np.random.default_rng(0)
    h = (rng.standard_normal((40, 14)) + 1j * rng.standard_normal((40, 14))).astype(np.complex64)
    assert np.array_equal(DO.denoise_plane(h, **zero), h)                      # zero weights: pure residual
    base = DO.denoise_plane(h, **w)
    h2 = h.copy()
    h2[20, 7] += 1.0
    changed = np.argwhere(DO.denoise_plane(h2, **w) != base)
    assert np.abs(changed - [20, 7]).max() <= 3                                 # receptive field of three 3x3 layers
    assert set(SHAPES) == set(w)


@pytest.mark.gpu
@pytest.mark.parametrize("n_sc,n_layers,n_items", [(72, 1, 3), (32, 1, 1), (1, 1, 2), (33, 2, 2), (300, 1, 2), (100, 4, 1), (3276, 1, 1)])
def test_hip_denoiser_matches_its_oracle(n_sc, n_layers, n_items):
    import torch
    from srsran_ce_pytorch_amd.denoiser import Denoiser
    w = random_weights(7, gain=0.7)
    rng = np.random.default_rng(n_sc)
    h = (rng.standard_normal((n_items, n_sc, 14, n_layers)) + 1j * rng.standard_normal((n_items, n_sc, 14, n_layers))).astype(np.complex64)
    h *= np.float32(0.8)
    want = DO.denoise(h, w)
    dn = Denoiser(w, "cuda:0")
    t = torch.from_numpy(h.copy()).cuda()
    assert dn(t) is t
    got = t.cpu().numpy()
    corr = np.abs(want - h).max()
    assert corr > 0.05                                                          # the network really does something
    assert np.abs(got - want).max() <= 2e-3 * np.abs(h).max(), (np.abs(got - want).max(), corr)


@pytest.mark.gpu
def test_hip_denoiser_zero_weights_and_errors():
    import torch
    from srsran_ce_pytorch_amd.denoiser import Denoiser
    w = {k: np.zeros(s, np.float32) for k, s in SHAPES.items()}
    t = torch.randn(2, 50, 14, 1, dtype=torch.complex64, device="cuda:0")
    ref = t.clone()
    assert torch.equal(Denoiser(w)(t), ref)                                     # exact identity
    with pytest.raises(NotImplementedError):
        Denoiser(w)(torch.zeros(1, 50, 12, 1, dtype=torch.complex64, device="cuda:0"))
    with pytest.raises(ValueError):
        Denoiser(w)(torch.zeros(1, 50, 14, 1, dtype=torch.complex128, device="cuda:0"))
    with pytest.raises(ValueError):
        Denoiser(dict(w, w2=np.zeros((16, 8, 3, 3), np.float32)))
    Denoiser(w)(torch.zeros(0, 50, 14, 1, dtype=torch.complex64, device="cuda:0"))    # empty batch: no launch


@pytest.mark.gpu
def test_denoiser_through_estimate_config_attribute():
    """`config.Denoiser` (optional attribute, read like the reference reads CNNSmoothingAlpha): estimate() = estimation
    followed by the in-place denoiser; the shim inherits it."""
    import torch
    from srsran_ce_pytorch_amd import estimator as E, synth as S
    from srsran_ce_pytorch_amd.denoiser import Denoiser
    case = S.case_spec("dn", 25, [S.hop_spec([2, 11], 2, 20)], seed=11)
    b = S.build_case(case, 2)
    g = torch.as_tensor(b.grids, device="cuda:0")[None]
    p = torch.as_tensor(b.pilots, device="cuda:0")
    plain = E.estimate(g, p, b.beta, b.hop1, b.hop2, b.config)
    w = random_weights(2)
    b.config.Denoiser = Denoiser(w)
    den = E.estimate(g, p, b.beta, b.hop1, b.hop2, b.config)
    want = DO.denoise(plain[0].cpu().numpy(), w)
    assert np.abs(den[0].cpu().numpy() - want).max() <= 2e-3 * np.abs(want).max()
    for a, c in zip(plain[1:], den[1:]):
        assert torch.equal(a, c)                                                # scalars untouched
    one = E.srs_channel_estimator(g[0, 1], p, b.beta, b.hop1, b.hop2, b.config)
    assert torch.allclose(one[0], den[0][0, 1], atol=1e-6)
