"""Seeded synthetic PUSCH slots for fixtures, parity tests and the bench.

A *case* is a plain JSON-able dict (so golden fixtures can carry it) describing the slot
geometry; ``build_case`` turns it into numpy hop/estimator configs, QPSK pilots and received
grids ``beta * H[sc] * pilot * CFO-ramp + AWGN`` with a dominant channel tap so the
time-alignment arg-max of the reference (`src/ce_rule_tensorized.py:686-696`) is tie-free.
The generator is the build's own code; nothing here comes from the
reference.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import numpy as np

TYPE1_CDM0 = [1, 0] * 6                      # DM-RS configuration type 1, CDM group 0
TYPE1_CDM1 = [0, 1] * 6                      # DM-RS configuration type 1, CDM group 1
TYPE2_CDM0 = [1, 1, 0, 0, 0, 0] * 2          # type 2: 4 pilots / PRB
TYPE2_CDM1 = [0, 0, 1, 1, 0, 0] * 2


def normal_cp_ms(scs_hz: float, n_syms: int = 14) -> np.ndarray:
    """Normal-CP durations in ms on the 2048-point numerology the reference harness assumes
    (`scripts/validation/validate_all.py:269-283`)."""
    scale = 15000.0 / scs_hz
    cp = np.array([round(160 * scale)] + [round(144 * scale)] * (n_syms - 1), dtype=np.float64)
    return cp * (1.0 / (scs_hz * 2048)) * 1000.0


def hop_spec(dmrs_symbols: List[int], prb_start: int, n_prbs: int, start_symbol: int = 0,
             n_alloc: int = 14, re_masks: Optional[List[List[int]]] = None,
             mask_prbs: Optional[List[int]] = None) -> Dict[str, Any]:
    """``mask_prbs``: PRB indices carrying DM-RS when they are not the contiguous run ``prb_start .. +n_prbs`` (the reference
    extracts pilots through ``maskPRBs`` but fills the grid through ``PRBstart`` / ``nPRBs``, T:571-576 vs T:301-304)."""
    h = dict(dmrs_symbols=list(dmrs_symbols), prb_start=int(prb_start), n_prbs=int(n_prbs),
             start_symbol=int(start_symbol), n_alloc=int(n_alloc),
             re_masks=[list(m) for m in (re_masks or [TYPE1_CDM0])])
    if mask_prbs is not None:
        h["mask_prbs"] = sorted(int(q) for q in mask_prbs)
    return h


def case_spec(name: str, n_prb_grid: int, hops: List[Dict[str, Any]], n_layers: int = 1,
              smoothing: str = "filter", cfo_compensate: bool = True, scs: float = 30e3,
              beta: float = 1.4125, n_sym: int = 14, seed: int = 0, cfo_hz: float = 250.0,
              delay_ns: float = 200.0, noise_var: float = 0.005) -> Dict[str, Any]:
    return dict(name=name, n_prb_grid=int(n_prb_grid), hops=hops, n_layers=int(n_layers),
                smoothing=smoothing, cfo_compensate=bool(cfo_compensate), scs=float(scs),
                beta=float(beta), n_sym=int(n_sym), seed=int(seed), cfo_hz=float(cfo_hz),
                delay_ns=float(delay_ns), noise_var=float(noise_var))


def _hop_arrays(case: Dict[str, Any], h: Dict[str, Any]) -> SimpleNamespace:
    n_sym, n_prb_grid = case["n_sym"], case["n_prb_grid"]
    dm = np.zeros(n_sym, bool)
    dm[h["dmrs_symbols"]] = True
    mp = np.zeros(n_prb_grid, bool)
    if h.get("mask_prbs") is not None:
        mp[h["mask_prbs"]] = True
    else:
        mp[h["prb_start"]: h["prb_start"] + h["n_prbs"]] = True
    re_mask = np.array(h["re_masks"], bool).T.reshape(12, -1)
    return SimpleNamespace(DMRSsymbols=dm, DMRSREmask=re_mask, PRBstart=h["prb_start"], nPRBs=h["n_prbs"],
                           maskPRBs=mp, startSymbol=h["start_symbol"], nAllocatedSymbols=h["n_alloc"])


def empty_hop_arrays() -> SimpleNamespace:
    return SimpleNamespace(DMRSsymbols=np.zeros((0,), bool), DMRSREmask=np.zeros((12, 0), bool), PRBstart=0,
                           nPRBs=0, maskPRBs=np.zeros((0,), bool), startSymbol=0, nAllocatedSymbols=0)


def qpsk_pilots(rng: np.random.Generator, n_re: int, n_dmrs: int, n_layers: int) -> np.ndarray:
    """(n_re, n_dmrs, L) unit-modulus QPSK; odd layers carry the frequency-domain OCC (+,-,+,-..)
    of their CDM partner so the reference's pair-averaging de-spread (T:620-628) separates them."""
    base = np.exp(1j * (math.pi / 4 + (math.pi / 2) * rng.integers(0, 4, size=(n_re, n_dmrs, (n_layers + 1) // 2))))
    occ = np.where(np.arange(n_re) % 2 == 0, 1.0, -1.0)[:, None]
    out = np.empty((n_re, n_dmrs, n_layers), np.complex64)
    for l in range(n_layers):
        out[:, :, l] = base[:, :, l // 2] * (occ if l % 2 else 1.0)
    return out


def build_case(case: Dict[str, Any], n_items: int = 1) -> SimpleNamespace:
    """Returns numpy configs + ``pilots (n_re, n_dmrs_total, L)`` + ``grids (n_items, n_sc, n_sym)``.

    All items share the pilots (as the Rx ports of one slot do) and differ in channel tap
    phases, CFO and noise.
    """
    rng = np.random.default_rng(case["seed"])
    n_sc, n_sym, L = 12 * case["n_prb_grid"], case["n_sym"], case["n_layers"]
    scs = case["scs"]
    hops = [_hop_arrays(case, h) for h in case["hops"]]
    hop1 = hops[0]
    hop2 = hops[1] if len(hops) > 1 else empty_hop_arrays()
    cp_ms = normal_cp_ms(scs)
    cfg = SimpleNamespace(scs=scs, CyclicPrefixDurations=cp_ms, Smoothing=case["smoothing"],
                          CFOCompensate=case["cfo_compensate"])
    cpd = cp_ms * scs / 1000.0
    sst = np.cumsum(np.concatenate([[cpd[0]], cpd[1:14] + 1.0]))          # symbols
    n_re = int(hop1.nPRBs * hop1.DMRSREmask[:, 0].sum())
    n_dmrs_tot = sum(len(h["dmrs_symbols"]) for h in case["hops"])
    pilots = qpsk_pilots(rng, n_re, n_dmrs_tot, L)

    f_sc = np.arange(n_sc) * scs
    tap_delay = np.array([0.0, 100e-9, 300e-9]) + case["delay_ns"] * 1e-9
    tap_amp = np.array([1.0, 0.35, 0.2])
    grids = np.empty((n_items, n_sc, n_sym), np.complex64)
    for it in range(n_items):
        cfo_norm = (case["cfo_hz"] * (0.5 + rng.random())) / scs * (1 if rng.random() < 0.5 else -1)
        g = (rng.standard_normal((n_sc, n_sym)) + 1j * rng.standard_normal((n_sc, n_sym))) * math.sqrt(case["noise_var"] / 2)
        s_off = 0
        for h, ha in zip(case["hops"], hops):
            H = np.zeros((n_sc, L), complex)
            for l in range(L):
                ph = np.exp(1j * 2 * math.pi * rng.random(3))
                H[:, l] = (tap_amp * ph)[None, :] @ np.exp(-2j * math.pi * tap_delay[:, None] * f_sc[None, :])
            for c in range(ha.DMRSREmask.shape[1]):
                res = np.flatnonzero(np.kron(ha.maskPRBs, ha.DMRSREmask[:, c]))
                for si, sym in enumerate(h["dmrs_symbols"]):
                    acc = np.zeros(res.size, complex)
                    for l in range(2 * c, min(L, 2 * c + 2)):
                        acc += H[res, l] * pilots[:, s_off + si, l]
                    g[res, sym] += case["beta"] * acc * np.exp(2j * math.pi * sst[sym] * cfo_norm)
            s_off += len(h["dmrs_symbols"])
        grids[it] = g.astype(np.complex64)
    return SimpleNamespace(case=case, hop1=hop1, hop2=hop2, config=cfg, pilots=pilots, grids=grids,
                           beta=case["beta"])


# --------------------------------------------------------------------------------------------
# Named geometries (BASELINE.json configs)
# --------------------------------------------------------------------------------------------
def bench_case(smoothing: str = "filter", n_layers: int = 1, seed: int = 1234) -> Dict[str, Any]:
    """273 PRB, DM-RS symbols [2, 11], type-1 mask, full-band, SCS 30 kHz (BASELINE.json configs 2-4)."""
    masks = [TYPE1_CDM0] if n_layers <= 2 else [TYPE1_CDM0, TYPE1_CDM1]
    return case_spec(f"pusch273_{smoothing}_L{n_layers}", 273, [hop_spec([2, 11], 0, 273, re_masks=masks)],
                     n_layers=n_layers, smoothing=smoothing, seed=seed)


def config1_case(seed: int = 7) -> Dict[str, Any]:
    """BASELINE.json configs[0]: 25 PRB inside the 52-PRB grid of the srsRAN vectors, one DM-RS
    symbol, LS only (no smoothing); the CFO path returns "not estimated"."""
    return case_spec("cfg1_25prb_1dmrs_none", 52, [hop_spec([2], 10, 25)], smoothing="none", scs=15e3, seed=seed)


# --------------------------------------------------------------------------------------------
# On-device generator for the bench (same signal model, torch ops, inputs born in HBM)
# --------------------------------------------------------------------------------------------
def torch_inputs(case: Dict[str, Any], n_slots: int, n_ports: int, device, seed: int, chunk: int = 256):
    """Returns ``(rx, pilots)``: ``rx`` is a ``[slots, ports, n_sym, n_sc]`` complex64 buffer viewed as
    ``[slots, ports, n_sc, n_sym]`` (subcarrier-contiguous: the layout the kernel's pilot loads
    coalesce on), ``pilots`` is ``[slots, n_re, n_dmrs_total, L]`` (one DM-RS sequence per slot, shared
    by its Rx ports).  AWGN on every RE; pilots REs carry beta*H*pilot*CFO-ramp."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    n_sc, n_sym, L = 12 * case["n_prb_grid"], case["n_sym"], case["n_layers"]
    scs, beta = case["scs"], case["beta"]
    hops = [_hop_arrays(case, h) for h in case["hops"]]
    cp_ms = normal_cp_ms(scs)
    cpd = cp_ms * scs / 1000.0
    sst = torch.as_tensor(np.cumsum(np.concatenate([[cpd[0]], cpd[1:14] + 1.0])), device=device, dtype=torch.float32)
    n_re = int(hops[0].nPRBs * hops[0].DMRSREmask[:, 0].sum())
    n_dmrs_tot = sum(len(h["dmrs_symbols"]) for h in case["hops"])

    k = torch.randint(0, 4, (n_slots, n_re, n_dmrs_tot, (L + 1) // 2), generator=g, device=device)
    base = torch.polar(torch.ones((), device=device), (math.pi / 4 + (math.pi / 2) * k).float())
    occ = torch.where(torch.arange(n_re, device=device) % 2 == 0, 1.0, -1.0)[None, :, None]
    pilots = torch.empty((n_slots, n_re, n_dmrs_tot, L), dtype=torch.complex64, device=device)
    for l in range(L):
        pilots[..., l] = base[..., l // 2] * (occ if l % 2 else 1.0)

    rx = torch.empty((n_slots, n_ports, n_sym, n_sc), dtype=torch.complex64, device=device)
    f_sc = torch.arange(n_sc, device=device, dtype=torch.float32) * scs
    tap_delay = torch.tensor([0.0, 100e-9, 300e-9], device=device) + case["delay_ns"] * 1e-9
    tap_amp = torch.tensor([1.0, 0.35, 0.2], device=device)
    steer = torch.polar(tap_amp[:, None].expand(3, n_sc).contiguous(), -2 * math.pi * tap_delay[:, None] * f_sc[None, :])
    sigma = math.sqrt(case["noise_var"] / 2)
    for b0 in range(0, n_slots, chunk):
        b1 = min(n_slots, b0 + chunk)
        nb = b1 - b0
        rxc = rx[b0:b1]
        rxc.copy_(torch.view_as_complex(torch.randn((nb, n_ports, n_sym, n_sc, 2), generator=g, device=device) * sigma))
        cfo = (case["cfo_hz"] * (0.5 + torch.rand((nb, n_ports), generator=g, device=device)) / scs) * \
            torch.where(torch.rand((nb, n_ports), generator=g, device=device) < 0.5, 1.0, -1.0)
        s_off = 0
        for h, ha in zip(case["hops"], hops):
            ph = torch.polar(torch.ones((), device=device), 2 * math.pi * torch.rand((nb, n_ports, L, 3), generator=g, device=device))
            H = torch.einsum("brlt,tk->brlk", ph, steer)                       # [nb, R, L, n_sc]
            for c in range(ha.DMRSREmask.shape[1]):
                res = torch.as_tensor(np.flatnonzero(np.kron(ha.maskPRBs, ha.DMRSREmask[:, c])), device=device)
                for si, sym in enumerate(h["dmrs_symbols"]):
                    acc = torch.zeros((nb, n_ports, res.numel()), dtype=torch.complex64, device=device)
                    for l in range(2 * c, min(L, 2 * c + 2)):
                        acc += H[:, :, l, res] * pilots[b0:b1, None, :, s_off + si, l]
                    ramp = torch.polar(torch.ones((), device=device), 2 * math.pi * sst[sym] * cfo)[:, :, None]
                    rxc[:, :, sym, res] += beta * acc * ramp
            s_off += len(h["dmrs_symbols"])
    return rx.permute(0, 1, 3, 2), pilots


def numpy_hops(case: Dict[str, Any]):
    hops = [_hop_arrays(case, h) for h in case["hops"]]
    cfg = SimpleNamespace(scs=case["scs"], CyclicPrefixDurations=normal_cp_ms(case["scs"]), Smoothing=case["smoothing"],
                          CFOCompensate=case["cfo_compensate"])
    return hops[0], (hops[1] if len(hops) > 1 else empty_hop_arrays()), cfg
