"""CPU restatement of the reference's LOOP-STYLE estimator, ``src/ce_rule_baseline.py`` ("B") -- TEST / BENCH INFRASTRUCTURE.

``ce_rule_baseline.py`` is the reference's MATLAB-shaped form of the same algorithm as ``ce_rule_tensorized.py``:
instead of one-shot tensor expressions it walks the pilot gaps, the layers and the CDM groups in Python loops.
BASELINE.json names it as the CPU baseline the GPU path is reported next to, so this file restates THAT loop structure
in numpy (one small array op per loop trip, as the reference issues one small torch op per trip):

* ``fill_ch_est_cdm``   B:237-360  full-grid clone per call (B:265), NaN-initialised hop band (B:281-283), one
                        interpolation per gap between consecutive pilots (B:303-320: ``prev + span * (1:stride)/(stride+1)``),
                        per-layer edge hold and grid write (B:333-358)
* ``compensate_cfo``    B:363-463  per-layer inner products summed pair-wise before the angle (B:415-428)
* ``process_hop``       B:507-758  per-CDM-group pilot extraction (B:583-605), per-layer smoothing (B:661-678), per-layer
                        reconstruction of the received pilots (B:727-739), grid fill per CDM group (B:742-744)
* ``srs_channel_estimator`` B:761-953 orchestration / normalisation / final CFO ramp, identical to T:745-937

The numerically delicate helpers that B shares verbatim with T (raised-cosine taps, virtual pilots, the float64 "same"
convolution, symbol start times) are imported from ``ce_oracle`` -- they are pinned there against the real reference.
Pinning of THIS file: ``tests/test_oracle_vs_golden.py::test_loop_baseline_matches_reference_fixture`` compares it with
every ``ce_rule_tensorized`` fixture (the reference's own baseline-vs-tensorized agreement on the same inputs, recorded in
``tests/golden/MANIFEST.json``, is <= 4.2e-8).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np

import ce_oracle as O

NRE = 12


def fill_ch_est_cdm(channel_in: np.ndarray, estimated: np.ndarray, hop, i_cdm: int) -> np.ndarray:
    """B:237-360 (``i_cdm`` 1-based as in the reference)."""
    channel_out = channel_in.copy()                                      # B:265: the whole grid is cloned on every call
    n_layers = estimated.shape[1]
    n_sc_hop = int(hop.nPRBs) * NRE
    estimated_all = np.full((n_sc_hop, n_layers), np.nan + 1j * np.nan, channel_out.dtype)   # B:281-283
    mask_all = np.tile(np.asarray(hop.DMRSREmask, bool)[:, i_cdm - 1], int(hop.nPRBs))       # B:287-290
    estimated_all[mask_all, :] = estimated
    filled = np.flatnonzero(mask_all).tolist()
    if not filled:
        return channel_out
    real_t = channel_out.real.dtype
    for i in range(len(filled) - 1):                                     # B:303-320: one interpolation per pilot gap
        start, stop = filled[i] + 1, filled[i + 1] - 1
        stride = stop - start + 1
        if stride <= 0:
            continue
        prev, nxt = estimated_all[start - 1, :], estimated_all[stop + 1, :]
        span = nxt - prev
        w = (np.arange(1, stride + 1, dtype=np.float64) / float(stride + 1)).astype(real_t)
        estimated_all[start:stop + 1, :] = prev[None, :] + w[:, None] * span[None, :]
    sc0 = NRE * int(hop.PRBstart)
    s0, s1 = int(hop.startSymbol), int(hop.startSymbol) + int(hop.nAllocatedSymbols)
    for il in range(n_layers):                                           # B:333-358
        estimated_all[filled[-1]:, il] = estimated_all[filled[-1], il]
        estimated_all[:filled[0] + 1, il] = estimated_all[filled[0], il]
        channel_out[sc0:sc0 + n_sc_hop, s0:s1, il + (i_cdm - 1) * 2] = estimated_all[:, il:il + 1]
    return channel_out


def compensate_cfo(rec_x: np.ndarray, dmrs_symbols: np.ndarray, scs_khz: float, cp_ms: np.ndarray, cfo_compensate: bool):
    """B:363-463.  Returns (rec_x_out, cfo | None)."""
    dmrs_ix = np.flatnonzero(np.asarray(dmrs_symbols, bool))
    if dmrs_ix.size < 2:
        return rec_x, None                                               # B:392-396
    n_layers = rec_x.shape[2]
    cpd = np.asarray(cp_ms, np.float64) * float(scs_khz)

    def inner(l0):                                                       # B:415-418
        return np.sum(np.conj(rec_x[:, 0, l0]) * rec_x[:, 1, l0], dtype=rec_x.dtype)

    acc = 0.0
    for l0 in range(0, n_layers - 1, 2):                                 # B:421-423: CDM pairs
        acc += float(np.angle(rec_x.dtype.type(inner(l0) + inner(l0 + 1))))
    if n_layers % 2 == 1:                                                # B:426-428
        acc += float(np.angle(inner(n_layers - 1)))
    n_samples = float(dmrs_ix[1] - dmrs_ix[0]) + float(np.sum(cpd[dmrs_ix[0] + 1:dmrs_ix[1] + 1]))
    cfo = acc / (2.0 * math.pi * n_samples) / float(math.ceil(n_layers / 2))
    if not cfo_compensate:
        return rec_x, cfo
    if cpd.size < 14:
        raise ValueError("cyclic_prefix_durations must have length >= 14 to match MATLAB code.")
    sst = np.cumsum(np.concatenate([[cpd[0]], cpd[1:14] + 1.0]))
    rot = np.exp(-1j * (2.0 * math.pi * sst * cfo)[dmrs_ix]).astype(rec_x.dtype)
    out = rec_x.copy()
    for il in range(n_layers):                                           # B:440-460: layer by layer
        out[:, :, il] = rec_x[:, :, il] * rot[None, :]
    return out, cfo


def process_hop(hop, pilots: np.ndarray, smoothing: str, rg: np.ndarray, scs: float, cp_ms: np.ndarray,
                cfo_compensate: bool, beta: float, sst: np.ndarray, channel: np.ndarray):
    """B:507-758.  Returns (epre, cfo_hop | None, ta, noise, rsrp, channel_out)."""
    pilots = O._c64(pilots)
    n_re, n_dmrs, n_layers = pilots.shape
    n_cdm = int(math.ceil(n_layers / 2))
    mask_prbs = np.asarray(hop.maskPRBs, bool)
    dmrs_mask = np.asarray(hop.DMRSsymbols, bool)
    dmrs_ix = np.flatnonzero(dmrs_mask)
    re_mask = np.asarray(hop.DMRSREmask, bool)
    beta32 = np.float32(beta)

    rx_pilots = np.zeros((n_re, n_dmrs, n_cdm), pilots.dtype)
    rec_x = np.zeros_like(pilots)
    epre = np.float64(0.0)
    mask_res = None
    for i_cdm in range(1, n_cdm + 1):                                    # B:583-605
        mask_res = np.kron(mask_prbs, re_mask[:, i_cdm - 1]).astype(bool)
        rx_sel = rg[mask_res][:, dmrs_ix]
        rx_pilots[:, :, i_cdm - 1] = rx_sel
        epre += O._abs2_sum(rx_sel)
        for il in range((i_cdm - 1) * 2, min(n_layers, i_cdm * 2)):
            rec_x[:, :, il] = rx_sel * np.conj(pilots[:, :, il])

    rec_nocfo, cfo_hop = compensate_cfo(rec_x, dmrs_mask, scs / 1000.0, cp_ms, cfo_compensate)
    p = (np.sum(rec_nocfo, axis=1, dtype=rec_nocfo.dtype) / beta32 / np.float32(n_dmrs)).astype(pilots.dtype)   # B:620
    if n_layers >= 2:                                                    # B:627-637
        m = min(p[0::2].shape[0], p[1::2].shape[0])
        if m:
            avg = (p[0:2 * m:2] + p[1:2 * m:2]) / np.float32(2)
            p[0:2 * m:2] = avg
            p[1:2 * m:2] = avg

    if smoothing == "mean":                                              # B:642-645
        p = (np.ones_like(p) * np.mean(p, axis=0, keepdims=True, dtype=p.dtype)).astype(p.dtype)
    elif smoothing == "filter":                                          # B:646-678
        dmrs_per_prb = int(re_mask[:, 0].sum())
        n_prb_active = int(mask_prbs.sum())
        rc = O.get_rc_filter(12 // dmrs_per_prb, min(3, n_prb_active))
        n_pils = min(12, rc.size // 2) if n_prb_active > 1 else dmrs_per_prb
        for il in range(n_layers):
            p[:, il] = O.smooth_filter_column(p[:, il].copy(), rc, n_pils)
    elif smoothing != "none":
        raise ValueError(f"Unknown smoothing strategy {smoothing}.")

    sc_resp = np.zeros((mask_res.size, n_layers), pilots.dtype)          # B:684-711
    sc_resp[mask_res] = p
    ir = np.fft.ifft(sc_resp, n=O.FFT_SIZE, axis=0).astype(pilots.dtype)
    a = np.abs(ir)
    power = np.sum(a * a, axis=1, dtype=a.dtype)
    head, tail = power[:O.HALF_CP], power[-O.HALF_CP:]
    i_delay, i_adv = int(np.argmax(head)), int(np.argmax(tail))
    i_max = i_delay if float(head[i_delay]) >= float(tail[i_adv]) else -(O.HALF_CP - (i_adv + 1) + 1)
    ta = float(i_max) / float(O.FFT_SIZE) / float(scs)

    est_rx = np.zeros_like(rx_pilots)                                    # B:714-744
    for i_cdm in range(1, n_cdm + 1):
        lo, hi = (i_cdm - 1) * 2, min(i_cdm * 2, n_layers)
        if cfo_compensate and cfo_hop is not None:
            ph = np.exp(1j * (2.0 * math.pi * sst * cfo_hop)[dmrs_mask]).astype(pilots.dtype)
            for il in range(lo, hi):                                     # B:727-733: one layer at a time
                hsym = p[:, il:il + 1] * ph[None, :]
                est_rx[:, :, i_cdm - 1] = est_rx[:, :, i_cdm - 1] + beta32 * pilots[:, :, il] * hsym
        else:
            for il in range(lo, hi):                                     # B:735-739
                hsym = np.broadcast_to(p[:, il:il + 1], (n_re, n_dmrs))
                est_rx[:, :, i_cdm - 1] = est_rx[:, :, i_cdm - 1] + beta32 * pilots[:, :, il] * hsym
        channel = fill_ch_est_cdm(channel, p[:, lo:hi], hop, i_cdm)
    noise = O._abs2_sum(rx_pilots - est_rx)
    rsrp = (np.float64(beta) ** 2) * O._abs2_sum(p) * float(n_dmrs)
    return epre, cfo_hop, ta, noise, rsrp, channel


def srs_channel_estimator(received_rg, pilots, beta_dmrs, hop1, hop2, config):
    """B:761-953: same boundary and return convention as ``ce_oracle.srs_channel_estimator`` (cfo_hz None = not estimated)."""
    rg = O._c64(received_rg)
    pilots = O._c64(pilots)
    n_layers = pilots.shape[2]
    cfo_compensate = bool(getattr(config, "CFOCompensate", True))
    smoothing = str(config.Smoothing) if getattr(config, "Smoothing", None) is not None else "filter"
    scs = float(config.scs)
    cp_ms = np.asarray(config.CyclicPrefixDurations, np.float64)
    sst = O.symbol_start_time(cp_ms, scs) if cfo_compensate else np.zeros((0,))
    channel = np.zeros((rg.shape[0], rg.shape[1], n_layers), rg.dtype)
    n1 = int(np.asarray(hop1.DMRSsymbols, bool).sum())
    epre, cfo, ta, noise, rsrp, channel = process_hop(hop1, pilots[:, :n1, :], smoothing, rg, scs, cp_ms, cfo_compensate,
                                                      float(beta_dmrs), sst, channel)
    all_dmrs = np.asarray(hop1.DMRSsymbols, bool).copy()
    h2 = np.asarray(hop2.DMRSsymbols)
    has_hop2 = h2.size != 0 and int(h2.astype(np.int64).sum()) != 0
    if has_hop2:
        h2 = h2.astype(bool)
        assert not np.any(all_dmrs & h2), "Hops should not overlap."
        all_dmrs |= h2
        assert np.array_equal(np.asarray(hop1.DMRSREmask), np.asarray(hop2.DMRSREmask)), \
            "The DM-RS mask should be the same for the two hops."
        e2, c2, t2, n2, r2, channel = process_hop(hop2, pilots[:, n1:, :], smoothing, rg, scs, cp_ms, cfo_compensate,
                                                  float(beta_dmrs), sst, channel)
        epre, ta, noise, rsrp = epre + e2, ta + t2, noise + n2, rsrp + r2
        if c2 is not None:
            cfo = (cfo + c2) / 2 if cfo is not None else c2
    n_pilots = int(hop1.nPRBs) * int(np.asarray(hop1.DMRSREmask, bool)[:, 0].sum()) * int(all_dmrs.sum())
    rsrp = rsrp / float(n_pilots) / float(n_layers)
    epre = epre / float(n_pilots)
    noise = noise / float(math.ceil(n_layers / 2) * n_pilots - 1)
    if has_hop2:
        ta = ta / 2.0
    if cfo_compensate and cfo is not None:
        rot = np.exp(1j * (2.0 * math.pi * sst * cfo)).astype(channel.dtype)
        channel = channel * rot[None, :, None]
    cfo_hz = None if cfo is None else cfo * scs
    return channel, np.float64(noise), np.float64(rsrp), np.float64(epre), np.float64(ta), cfo_hz
