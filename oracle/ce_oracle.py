"""CPU oracle for the PUSCH DM-RS channel-estimation hot path (TEST INFRASTRUCTURE ONLY).

This file is a numpy restatement of the algorithm the reference implements in
``src/ce_rule_tensorized.py`` ("T" below; ``src/ce_rule_baseline.py`` is the same
arithmetic written with Python loops and ``src/ce_dl_cnn.py`` swaps the frequency
interpolation for a fixed 3-tap in-painting stencil, restated in ``ce_oracle_cnn``
helpers at the bottom of this file).  One call = one slot x one Rx port.

It exists to *check* the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``srsran_ce_pytorch_amd``) never does and has no CPU fallback.

Parity pinning: the reference is importable in the build container, so this oracle is
pinned by ``tests/golden/*.npz`` -- inputs plus the six outputs of the real reference,
produced by ``tools/make_golden.py`` (committed).  ``tests/test_oracle_vs_golden.py``
checks every fixture.  The srsRAN/MATLAB ``.dat`` vectors the reference's scripts read are
git-ignored upstream and absent, so "vs MATLAB vectors" is pinned only transitively.

dtype conventions follow the reference: grid math in complex64/float32, the scalar
accumulators / regression / RC taps / phasor angles in float64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

NRE = 12          # subcarriers per PRB
FFT_SIZE = 4096   # time-alignment IFFT length (T:677)
HALF_CP = int(math.floor((144 / 2) * FFT_SIZE / 2048))  # 144 bins each side (T:682)


# --------------------------------------------------------------------------------------
# Configuration containers (field names follow T:13-29)
# --------------------------------------------------------------------------------------
@dataclass
class HopConfig:
    DMRSsymbols: np.ndarray      # (n_sym,) bool
    DMRSREmask: np.ndarray       # (12, nCDM) bool
    PRBstart: int
    nPRBs: int
    maskPRBs: np.ndarray         # (n_prb_grid,) bool
    startSymbol: int
    nAllocatedSymbols: int


@dataclass
class EstimatorConfig:
    scs: float
    CyclicPrefixDurations: np.ndarray   # (>=14,) milliseconds
    Smoothing: str = "filter"
    CFOCompensate: bool = True
    CNNSmoothingAlpha: float = 0.0      # only read by the "cnn" interpolation (C:864)


def empty_hop() -> HopConfig:
    """Hop with no DM-RS: how the harness encodes "no second hop" (VA:449-457)."""
    return HopConfig(np.zeros((0,), bool), np.zeros((12, 0), bool), 0, 0, np.zeros((0,), bool), 0, 0)


# --------------------------------------------------------------------------------------
# Small helpers
# --------------------------------------------------------------------------------------
def _c64(x) -> np.ndarray:
    x = np.asarray(x)
    return x if np.iscomplexobj(x) else x.astype(np.complex64)


def _abs2_sum(x: np.ndarray) -> np.float64:
    """||x||_F^2 computed the way T:454-456 does: |x| (float32) squared, float32 sum."""
    a = np.abs(x)
    return np.float64(np.sum(a * a, dtype=a.dtype))


def symbol_start_time(cp_ms: np.ndarray, scs_hz: float) -> np.ndarray:
    """cumsum([CPD0, CPD1..13 + 1]) with CPD = cp_ms*scs/1000, in units of symbols (T:809-820)."""
    cpd = np.asarray(cp_ms, np.float64) * float(scs_hz) / 1000.0
    if cpd.size < 14:
        raise ValueError("config.CyclicPrefixDurations must have length >= 14 to match MATLAB code.")
    v = np.empty(14, np.float64)
    v[0] = cpd[0]
    v[1:] = cpd[1:14] + 1.0
    return np.cumsum(v)


def unwrap_1d(ph: np.ndarray) -> np.ndarray:
    """Phase unwrap with numpy.unwrap's +pi convention (T:35-66)."""
    if ph.size <= 1:
        return ph
    dd = np.diff(ph)
    ddmod = np.remainder(dd + math.pi, 2.0 * math.pi) - math.pi
    ddmod = np.where((ddmod == -math.pi) & (dd > 0), ddmod + 2.0 * math.pi, ddmod)
    corr = np.where(np.abs(dd) < math.pi, 0.0, ddmod - dd)
    return ph + np.concatenate([[0.0], np.cumsum(corr)])


def create_virtual_pilots(in_pilots: np.ndarray, n_virtuals: int) -> np.ndarray:
    """Straight-line fit of |.| and of unwrapped angle, extrapolated to -nV..-1 (T:69-140)."""
    if n_virtuals < 0:
        raise ValueError("n_virtuals must be >= 0")
    in_pilots = _c64(in_pilots)
    if n_virtuals == 0:
        return np.empty((0,), np.complex64)
    n = in_pilots.size
    if n == 0:
        raise ValueError("in_pilots must be non-empty")
    if n == 1:  # T:95-101 constant hold
        amp = np.repeat(np.abs(in_pilots), n_virtuals)
        ph = np.repeat(np.angle(in_pilots), n_virtuals)
        return (amp * np.exp(1j * ph)).astype(in_pilots.dtype)
    x = np.arange(n, dtype=np.float64)
    mx = x.mean()
    denom = np.sum(x * x) - n * mx * mx
    k = np.arange(-n_virtuals, 0, dtype=np.float64)

    def line(y):
        my = y.mean()
        a = (np.sum(x * y) - n * mx * my) / denom
        return a * k + (my - a * mx)

    amp = line(np.abs(in_pilots).astype(np.float64))
    ph = line(unwrap_1d(np.angle(in_pilots).astype(np.float64)))
    return (amp * np.exp(1j * ph)).astype(in_pilots.dtype)


def rcosdesign_normal(beta: float, span: int, sps: int) -> np.ndarray:
    """Raised-cosine impulse response on t = -span/2 : 1/sps : span/2 (T:143-181)."""
    n = np.arange(-span * sps // 2, span * sps // 2 + 1, dtype=np.float64)
    t = n / float(sps)
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc = np.where(t == 0, 1.0, np.sin(math.pi * t) / (math.pi * t))
        h = sinc * np.cos(math.pi * beta * t) / (1.0 - (2.0 * beta * t) ** 2)
    if beta > 0:
        t0 = 1.0 / (2.0 * beta)
        sing = ~np.isfinite(h) | (np.abs(np.abs(t) - t0) < (1.0 / sps) * 1e-6)
        h = np.where(sing, (math.pi * beta / 2.0) * math.sin(1.0 / (2.0 * beta)), h)
    return h


def get_rc_filter(stride: int, n_rbs: int) -> np.ndarray:
    """RC taps (roll-off 0.2, 10 samples/RB) decimated by ``stride``, unit sum (T:184-234).

    The reference also returns a ``correction`` vector that no caller uses (T:642); omitted.
    """
    if stride <= 0:
        raise ValueError("stride must be >= 1")
    if n_rbs <= 0:
        raise ValueError("n_rbs must be >= 1")
    ff = rcosdesign_normal(0.2, n_rbs, 10)
    half = ff.size // 2
    kmax = (half // stride) * stride
    taps = ff[np.arange(-kmax, kmax + 1, stride) + (ff.size - 1) // 2]
    return taps / taps.sum()


def conv_same_real_taps(x: np.ndarray, h: np.ndarray) -> np.ndarray:
    """conv(x, h, 'same') with zero padding len(h)//2, float64 MACs, cast back (T:459-493)."""
    x = _c64(x)
    pad = h.size // 2
    full_r = np.convolve(x.real.astype(np.float64), h.astype(np.float64))
    full_i = np.convolve(x.imag.astype(np.float64), h.astype(np.float64))
    out_len = x.size + 2 * pad - h.size + 1
    y = full_r[h.size - 1 - pad: h.size - 1 - pad + out_len] + 1j * full_i[h.size - 1 - pad: h.size - 1 - pad + out_len]
    return y.astype(x.dtype)


# --------------------------------------------------------------------------------------
# CFO estimation / compensation (T:357-451)
# --------------------------------------------------------------------------------------
def compensate_cfo(rec_x: np.ndarray, dmrs_symbols: np.ndarray, scs_khz: float,
                   cp_ms: np.ndarray, cfo_compensate: bool):
    """Returns (rec_x_out, cfo) with cfo=None when the hop has < 2 DM-RS symbols."""
    dmrs_ix = np.flatnonzero(np.asarray(dmrs_symbols, bool))
    if dmrs_ix.size < 2:
        return rec_x, None
    n_layers = rec_x.shape[2]
    cpd = np.asarray(cp_ms, np.float64) * float(scs_khz)
    inner = np.sum(np.conj(rec_x[:, 0, :]) * rec_x[:, 1, :], axis=0, dtype=rec_x.dtype)  # (L,)
    acc = 0.0
    n_even = (n_layers // 2) * 2
    if n_even:
        acc += float(np.sum(np.angle(inner[:n_even].reshape(-1, 2).sum(axis=1, dtype=inner.dtype)).astype(np.float64)))
    if n_layers % 2:
        acc += float(np.angle(inner[-1]))
    n_samples = float(dmrs_ix[1] - dmrs_ix[0]) + float(np.sum(cpd[dmrs_ix[0] + 1: dmrs_ix[1] + 1]))
    cfo = acc / (2.0 * math.pi * n_samples) / float(math.ceil(n_layers / 2))
    if not cfo_compensate:
        return rec_x, cfo
    if cpd.size < 14:
        raise ValueError("cyclic_prefix_durations must have length >= 14 to match MATLAB code.")
    v = np.empty(14, np.float64)
    v[0] = cpd[0]
    v[1:] = cpd[1:14] + 1.0
    sst = np.cumsum(v)
    rot = np.exp(-1j * (2.0 * math.pi * sst * cfo)[dmrs_ix]).astype(rec_x.dtype)
    return rec_x * rot[None, :, None], cfo


# --------------------------------------------------------------------------------------
# Frequency interpolation + grid fill (T:237-354)
# --------------------------------------------------------------------------------------
def interp_linear(estimated: np.ndarray, mask_all: np.ndarray) -> np.ndarray:
    """Linear interpolation between pilot REs, flat hold outside the first/last pilot."""
    n_sc_hop = mask_all.size
    filled = np.flatnonzero(mask_all)
    pos = np.arange(n_sc_hop)
    out = np.empty((n_sc_hop, estimated.shape[1]), estimated.dtype)
    first, last = filled[0], filled[-1]
    out[pos <= first] = estimated[0]
    out[pos >= last] = estimated[-1]
    mid = (pos > first) & (pos < last)
    if mid.any():
        r_ord = np.searchsorted(filled, pos[mid], side="left")
        l_ord = r_ord - 1
        lp, rp = filled[l_ord], filled[r_ord]
        real_t = estimated.real.dtype
        alpha = ((pos[mid].astype(real_t) - lp.astype(real_t)) / (rp.astype(real_t) - lp.astype(real_t)))[:, None]
        lv, rv = estimated[l_ord], estimated[r_ord]
        out[mid] = lv + alpha * (rv - lv)
    return out


def fill_ch_est_cdm(channel: np.ndarray, estimated: np.ndarray, hop: HopConfig, i_cdm0: int,
                    interp: str = "linear") -> None:
    """Interpolate ``estimated`` (n_re, Lc) over the hop band and write it, replicated over the
    hop's symbols, into ``channel`` in place (the reference clones; callers here own the grid)."""
    mask_all = np.tile(np.asarray(hop.DMRSREmask, bool)[:, i_cdm0], int(hop.nPRBs))
    if not mask_all.any():
        return
    if interp == "linear":
        est_all = interp_linear(estimated, mask_all)
    elif interp == "cnn":
        est_all = interp_cnn(estimated, mask_all)
    else:
        raise ValueError(f"unknown interp {interp}")
    sc0 = NRE * int(hop.PRBstart)
    s0 = int(hop.startSymbol)
    for il in range(estimated.shape[1]):
        channel[sc0: sc0 + mask_all.size, s0: s0 + int(hop.nAllocatedSymbols), il + 2 * i_cdm0] = est_all[:, il:il + 1]


# --------------------------------------------------------------------------------------
# EXTENSION (no counterpart in the reference -> "parity unpinned"): block-wise LMMSE / Wiener frequency
# smoothing, Smoothing="mmse".  This restatement is the only oracle the GPU path has for it.
#   model     y_k = h_k + n_k at the pilot REs; channel taps uniformly spread over [0, tau]:
#             r(d) = E[h(f+d) h(f)*] = sinc(d*tau) * exp(-j*pi*d*tau)   (d in Hz, r(0) = 1)
#   filter    W = R (R + nsr*I)^-1 over blocks of MMSE_BLOCK consecutive pilots (the last block is anchored at
#             the band end and only supplies the outputs the previous blocks did not); nsr = noise-to-signal ratio
# --------------------------------------------------------------------------------------
MMSE_BLOCK = 32


def mmse_matrix(pilot_sc: np.ndarray, scs_hz: float, tau_s: float, nsr: float) -> np.ndarray:
    """W (M x M, complex128) for pilots at subcarrier indices ``pilot_sc`` (one block)."""
    d = (pilot_sc[:, None] - pilot_sc[None, :]).astype(np.float64) * float(scs_hz)
    r = np.sinc(d * tau_s) * np.exp(-1j * math.pi * d * tau_s)
    return r @ np.linalg.inv(r + nsr * np.eye(pilot_sc.size))


def smooth_mmse(col: np.ndarray, pilot_sc: np.ndarray, scs_hz: float, tau_s: float, nsr: float) -> np.ndarray:
    n = col.size
    m = min(MMSE_BLOCK, n)
    w = mmse_matrix(pilot_sc[:m], scs_hz, tau_s, nsr).astype(np.complex64)
    nb = -(-n // m)
    out = np.empty_like(col)
    for b in range(nb):
        s0 = min(b * m, n - m)
        y = (w.astype(np.complex128) @ col[s0: s0 + m].astype(np.complex128)).astype(col.dtype)
        lo = b * m
        out[lo: s0 + m] = y[lo - s0:]
    return out


def mmse_params(config, cp_ms: np.ndarray):
    """(tau_s, nsr): MMSEDelaySpread defaults to the normal cyclic prefix (CyclicPrefixDurations[1], ms)."""
    tau = getattr(config, "MMSEDelaySpread", None)
    tau = float(cp_ms[1]) * 1e-3 if tau is None else float(tau)
    nsr = getattr(config, "MMSENoiseToSignal", None)
    return tau, (0.01 if nsr is None else float(nsr))


# --------------------------------------------------------------------------------------
# One hop (T:495-739) and the slot-level driver (T:745-937)
# --------------------------------------------------------------------------------------
def smooth_filter_column(col: np.ndarray, rc: np.ndarray, n_pils: int) -> np.ndarray:
    """Virtual pilots on both band edges + RC FIR + crop for one layer (T:649-664)."""
    v_begin = create_virtual_pilots(col[:n_pils], n_pils)
    v_end = create_virtual_pilots(col[-n_pils:][::-1], n_pils)
    x = np.concatenate([v_begin, col, v_end[::-1]])
    y = conv_same_real_taps(x, rc)
    return y[n_pils: y.size - n_pils]


def process_hop(hop: HopConfig, pilots: np.ndarray, smoothing: str, rg: np.ndarray, scs: float,
                cp_ms: np.ndarray, cfo_compensate: bool, beta: float, sst: np.ndarray,
                channel: np.ndarray, interp: str = "linear", cnn_alpha: float = 0.0, mmse_cfg=None, stages=None):
    """Returns (epre, cfo_hop|None, ta, noise, rsrp) contributions of this hop; fills ``channel``.
    ``stages`` (a list, tests only) receives a dict of this hop's intermediate results."""
    pilots = _c64(pilots)
    n_re, n_dmrs, n_layers = pilots.shape
    n_cdm = int(math.ceil(n_layers / 2))
    mask_prbs = np.asarray(hop.maskPRBs, bool)
    dmrs_mask = np.asarray(hop.DMRSsymbols, bool)
    dmrs_ix = np.flatnonzero(dmrs_mask)
    re_mask = np.asarray(hop.DMRSREmask, bool)

    rx_pilots = np.empty((n_re, n_dmrs, n_cdm), pilots.dtype)
    rec_x = np.empty_like(pilots)
    epre = np.float64(0.0)
    mask_res = None
    for c in range(n_cdm):                                            # S1-S3  T:571-593
        mask_res = np.kron(mask_prbs, re_mask[:, c]).astype(bool)
        rx_sel = rg[mask_res][:, dmrs_ix]
        rx_pilots[:, :, c] = rx_sel
        epre += _abs2_sum(rx_sel)
        lo, hi = 2 * c, min(n_layers, 2 * c + 2)
        rec_x[:, :, lo:hi] = rx_sel[:, :, None] * np.conj(pilots[:, :, lo:hi])

    rec_nocfo, cfo_hop = compensate_cfo(rec_x, dmrs_mask, scs / 1000.0, cp_ms, cfo_compensate)   # S4

    beta32 = np.float32(beta)
    p = (np.sum(rec_nocfo, axis=1, dtype=rec_nocfo.dtype) / beta32 / np.float32(n_dmrs)).astype(pilots.dtype)  # S5 T:613
    if n_layers >= 2:                                                 # S6  T:620-628
        m = min(p[0::2].shape[0], p[1::2].shape[0])
        if m:
            avg = (p[0:2 * m:2] + p[1:2 * m:2]) / np.float32(2)
            p[0:2 * m:2] = avg
            p[1:2 * m:2] = avg

    p_ls = p.copy()
    if smoothing == "mean":                                           # S7  T:633-668
        p = (np.ones_like(p) * np.mean(p, axis=0, keepdims=True, dtype=p.dtype)).astype(p.dtype)
    elif smoothing == "filter":
        dmrs_per_prb = int(re_mask[:, 0].sum())
        n_prb_active = int(mask_prbs.sum())
        rc = get_rc_filter(12 // dmrs_per_prb, min(3, n_prb_active))
        n_pils = min(12, rc.size // 2) if n_prb_active > 1 else dmrs_per_prb
        for il in range(n_layers):
            sm = smooth_filter_column(p[:, il].copy(), rc, n_pils)
            if interp == "cnn" and cnn_alpha > 0.0:                   # C:712-715
                a = float(max(0.0, min(1.0, cnn_alpha)))
                sm = (sm + a * (cnn_lowpass(sm, passes=1) - sm)).astype(p.dtype)
            p[:, il] = sm
    elif smoothing == "mmse":                                         # extension, see mmse_matrix()
        tau, nsr = mmse_cfg
        for il in range(n_layers):
            sc_l = np.flatnonzero(np.kron(mask_prbs, re_mask[:, il // 2]))
            p[:, il] = smooth_mmse(p[:, il].copy(), sc_l, scs, tau, nsr)
    elif smoothing != "none":
        raise ValueError(f"Unknown smoothing strategy {smoothing}.")

    # S8 time alignment: scatter with the LAST CDM group's RE mask for every layer (T:672-675)
    sc_resp = np.zeros((mask_res.size, n_layers), pilots.dtype)
    sc_resp[mask_res] = p
    ir = np.fft.ifft(sc_resp, n=FFT_SIZE, axis=0).astype(pilots.dtype)
    a = np.abs(ir)
    power = np.sum(a * a, axis=1, dtype=a.dtype)
    head, tail = power[:HALF_CP], power[-HALF_CP:]
    i_delay, i_adv = int(np.argmax(head)), int(np.argmax(tail))
    i_max = i_delay if float(head[i_delay]) >= float(tail[i_adv]) else -(HALF_CP - (i_adv + 1) + 1)
    ta = float(i_max) / float(FFT_SIZE) / float(scs)
    if stages is not None:
        # power at the chosen bin and at its two neighbours among the examined bins; bins 4095 (the advance side's last)
        # and 0 (the delay side's first) are neighbours too -- a peak between them is decided by the `>=` above
        if i_max >= 0:
            lo = float(head[i_delay - 1]) if i_delay > 0 else float(tail[HALF_CP - 1])
            top, hi = float(head[i_delay]), float(head[i_delay + 1]) if i_delay + 1 < HALF_CP else -1.0
        else:
            lo = float(tail[i_adv - 1]) if i_adv > 0 else -1.0
            top, hi = float(tail[i_adv]), float(tail[i_adv + 1]) if i_adv + 1 < HALF_CP else float(head[0])
        stages.append(dict(p_ls=p_ls, p_smooth=p.copy(), cfo_hop=cfo_hop, ta_bin=i_max, ta_pw=[lo, top, hi]))

    # S9-S11: reconstruct the received pilots, fill the grid, residual + RSRP (T:700-730)
    est_rx = np.zeros_like(rx_pilots)
    for c in range(n_cdm):
        lo, hi = 2 * c, min(n_layers, 2 * c + 2)
        h_cdm = p[:, lo:hi]
        if cfo_compensate and cfo_hop is not None:
            ph = np.exp(1j * (2.0 * math.pi * sst * cfo_hop)[dmrs_mask]).astype(pilots.dtype)
            hsym = h_cdm[:, None, :] * ph[None, :, None]
        else:
            hsym = np.broadcast_to(h_cdm[:, None, :], (n_re, n_dmrs, hi - lo))
        est_rx[:, :, c] = beta32 * np.sum(pilots[:, :, lo:hi] * hsym, axis=2, dtype=pilots.dtype)
        fill_ch_est_cdm(channel, h_cdm, hop, c, interp)
    noise = _abs2_sum(rx_pilots - est_rx)
    rsrp = (np.float64(beta) ** 2) * _abs2_sum(p) * float(n_dmrs)
    return epre, cfo_hop, ta, noise, rsrp


def srs_channel_estimator(received_rg, pilots, beta_dmrs, hop1: HopConfig, hop2: HopConfig,
                          config: EstimatorConfig, interp: str = "linear", stages=None):
    """Oracle for T:745-937 (``interp="cnn"`` selects the ce_dl_cnn.py fill, C:233-352).  ``stages``: an empty list that
    receives one dict per hop (LS estimate, smoothed estimate, hop CFO, TA bin and the power around it) -- tests only.

    Returns (channel_est_rg (n_sc,n_sym,L) complex64, noise, rsrp, epre, time_alignment,
    cfo_hz) with float64 scalars; cfo_hz is None where the reference returns an empty tensor.
    """
    rg = _c64(received_rg)
    pilots = _c64(pilots)
    n_layers = pilots.shape[2]
    cfo_compensate = bool(getattr(config, "CFOCompensate", True))
    smoothing = str(config.Smoothing) if getattr(config, "Smoothing", None) is not None else "filter"
    scs = float(config.scs)
    cp_ms = np.asarray(config.CyclicPrefixDurations, np.float64)
    cnn_alpha = float(getattr(config, "CNNSmoothingAlpha", 0.0) or 0.0)
    sst = symbol_start_time(cp_ms, scs) if cfo_compensate else np.zeros((0,))

    mmse_cfg = mmse_params(config, cp_ms) if smoothing == "mmse" else None
    channel = np.zeros((rg.shape[0], rg.shape[1], n_layers), rg.dtype)
    n1 = int(np.asarray(hop1.DMRSsymbols, bool).sum())
    epre, cfo, ta, noise, rsrp = process_hop(hop1, pilots[:, :n1, :], smoothing, rg, scs, cp_ms,
                                             cfo_compensate, float(beta_dmrs), sst, channel, interp, cnn_alpha, mmse_cfg, stages)
    all_dmrs = np.asarray(hop1.DMRSsymbols, bool).copy()
    h2 = np.asarray(hop2.DMRSsymbols)
    has_hop2 = h2.size != 0 and int(h2.astype(np.int64).sum()) != 0
    if has_hop2:
        h2 = h2.astype(bool)
        assert not np.any(all_dmrs & h2), "Hops should not overlap."
        all_dmrs |= h2
        assert np.array_equal(np.asarray(hop1.DMRSREmask), np.asarray(hop2.DMRSREmask)), \
            "The DM-RS mask should be the same for the two hops."
        e2, c2, t2, n2, r2 = process_hop(hop2, pilots[:, n1:, :], smoothing, rg, scs, cp_ms,
                                         cfo_compensate, float(beta_dmrs), sst, channel, interp, cnn_alpha, mmse_cfg, stages)
        epre, ta, noise, rsrp = epre + e2, ta + t2, noise + n2, rsrp + r2
        if c2 is not None:
            cfo = (cfo + c2) / 2 if cfo is not None else c2          # T:605-609

    n_pilots = int(hop1.nPRBs) * int(np.asarray(hop1.DMRSREmask, bool)[:, 0].sum()) * int(all_dmrs.sum())
    rsrp = rsrp / float(n_pilots) / float(n_layers)
    epre = epre / float(n_pilots)
    noise = noise / float(math.ceil(n_layers / 2) * n_pilots - 1)
    if has_hop2:
        ta = ta / 2.0
    if cfo_compensate and cfo is not None:                             # T:921-929
        rot = np.exp(1j * (2.0 * math.pi * sst * cfo)).astype(channel.dtype)
        channel = channel * rot[None, :, None]
    cfo_hz = None if cfo is None else cfo * scs
    return channel, np.float64(noise), np.float64(rsrp), np.float64(epre), np.float64(ta), cfo_hz


# --------------------------------------------------------------------------------------
# ce_dl_cnn.py's fixed-weight 1-D stencil (C:433-508) -- restated for the secondary path
# --------------------------------------------------------------------------------------
_H3 = np.array([0.25, 0.5, 0.25], np.float64)


def _conv3_reflect(x: np.ndarray) -> np.ndarray:
    """3-tap [.25,.5,.25] 'same' convolution, reflect padding (replicate for length 1) (C:433-451)."""
    x = np.asarray(x, np.float64)
    if x.size == 1:
        xp = np.concatenate([x, x, x])
    else:
        xp = np.concatenate([x[1:2], x, x[-2:-1]])
    return _H3[0] * xp[2:] + _H3[1] * xp[1:-1] + _H3[2] * xp[:-2]


def cnn_lowpass(x: np.ndarray, passes: int = 2) -> np.ndarray:
    """C:454-470."""
    x = _c64(x)
    if x.size <= 2:
        return x
    yr, yi = x.real.astype(np.float64), x.imag.astype(np.float64)
    for _ in range(max(1, int(passes))):
        yr, yi = _conv3_reflect(yr), _conv3_reflect(yi)
    return (yr + 1j * yi).astype(x.dtype)


def cnn_inpaint(x_sparse: np.ndarray, known: np.ndarray, n_iters: int) -> np.ndarray:
    """Partial-convolution in-painting, complex64 round trip per iteration (C:473-508)."""
    x_sparse = _c64(x_sparse)
    known = np.asarray(known, bool).ravel()
    if known.all():
        return cnn_lowpass(x_sparse, 2)
    x_known = x_sparse.copy()
    x = x_sparse.copy()
    m = known.astype(np.float64)
    eps = 1e-12
    for _ in range(max(1, int(n_iters))):
        den = _conv3_reflect(m)
        num_r = _conv3_reflect(x.real.astype(np.float64) * m)
        num_i = _conv3_reflect(x.imag.astype(np.float64) * m)
        prop = (num_r / (den + eps) + 1j * (num_i / (den + eps))).astype(x.dtype)
        m = np.maximum(m, (den > eps).astype(np.float64))
        x = np.where(known, x_known, prop)
    return np.where(known, x_known, cnn_lowpass(x, 2))


def interp_cnn(estimated: np.ndarray, mask_all: np.ndarray) -> np.ndarray:
    """C:276-295: seed pilots into zeros, in-paint each layer with max(6, n_sc_hop//8) iterations."""
    n_sc_hop = mask_all.size
    out = np.zeros((n_sc_hop, estimated.shape[1]), estimated.dtype)
    out[mask_all] = estimated
    n_iters = max(6, n_sc_hop // 8)
    for il in range(estimated.shape[1]):
        out[:, il] = cnn_inpaint(out[:, il], mask_all, n_iters)
    return out
