"""Host side of the estimator boundary: tensor-in / tensor-out, HIP kernels underneath.

Two entry points:

* ``estimate(received_rg[B,R,n_sc,n_sym], pilots, beta_dmrs, hop1, hop2, config)`` -- the batched
  form: every slot x Rx-port pair is one workgroup of one fused launch.
* ``srs_channel_estimator(received_rg[n_sc,n_sym], pilots, beta_dmrs, hop1, hop2, config)`` --
  signature, return types and error behaviour of the reference function
  (`src/ce_rule_tensorized.py:745-937`), so its validation scripts run unchanged
  (`scripts/validation/validate_all.py:541-545`).

PyTorch is used for device memory and streams only; all arithmetic happens in
``csrc/libce_hip.so`` (C ABI: ``include/ce_hip.h``).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
import warnings
from collections import OrderedDict
from typing import Any, Optional, Tuple

import numpy as np
import torch

from . import _lib


def _np(x, dtype=None) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    a = np.asarray(x)
    return a.astype(dtype) if dtype is not None else a


class Plan:
    """Immutable per-configuration state: descriptor resolved on the host, tables on the GPU."""

    def __init__(self, handle: int, info: _lib.PlanInfo, key, device: torch.device, n_layers: int, n_sym: int):
        self._handle = C.c_void_p(handle)
        self.key = key
        self.device = device
        self.n_layers = n_layers
        self.n_sym = n_sym
        self.n_sc = info.n_sc
        self.n_re = info.n_re
        self.n_dmrs_total = info.n_dmrs_total
        self.cfo_estimated = bool(info.cfo_estimated)
        self.lds_bytes = info.lds_bytes
        self.threads = info.threads
        self.alg_bytes_per_item = info.alg_bytes_per_item
        self.pilot_bytes_per_slot = info.pilot_bytes_per_slot

    def __del__(self):
        try:
            if self._handle:
                _lib.load().ce_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


_PLAN_CACHE: "OrderedDict[Any, Plan]" = OrderedDict()
_PLAN_CACHE_MAX = 64
_PLAN_CACHE_LOCK = threading.Lock()


def _hop_fields(hop, n_cdm: int):
    dm = _np(hop.DMRSsymbols).astype(bool).ravel()
    rm = _np(hop.DMRSREmask).astype(bool)
    mp = _np(hop.maskPRBs).astype(bool).ravel()
    return dm, rm, mp


def _raise_for(code: int):
    msg = _lib.last_error()
    if code == _lib.CE_ERR_INVALID:
        if msg in ("Hops should not overlap.", "The DM-RS mask should be the same for the two hops."):
            raise AssertionError(msg)                      # T:862, T:869
        raise ValueError(msg)
    if code == _lib.CE_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"libce_hip: {msg} (code {code})")


def _resolve(hop1, hop2, config, beta_dmrs, n_layers: int, n_prb_grid: int, n_sym: int, interp: str):
    """Validate like the reference and reduce (hop1, hop2, config) to plain values + the plan-cache key.  This is all a
    cache hit pays for; the ctypes descriptor is only built on a miss (``_fill_desc``).

    Raises what the reference raises for the same inputs: ``ValueError`` for an unknown smoothing strategy
    (T:668) or a cyclic-prefix vector shorter than 14 (T:816), ``AssertionError`` for overlapping hops /
    different DM-RS masks (T:862, T:869).
    """
    cfo_comp = bool(getattr(config, "CFOCompensate", True))
    smoothing = str(config.Smoothing) if getattr(config, "Smoothing", None) is not None else "filter"
    if smoothing not in _lib.SMOOTHING:
        raise ValueError(f"Unknown smoothing strategy {smoothing}.")
    if interp not in _lib.INTERP:
        raise ValueError(f"unknown interp {interp}")
    cp = _np(config.CyclicPrefixDurations, np.float64).ravel()
    if cfo_comp and cp.size < 14:
        raise ValueError("config.CyclicPrefixDurations must have length >= 14 to match MATLAB code.")
    cp14 = np.zeros(14, np.float64)
    cp14[: min(14, cp.size)] = cp[:14]
    scs = float(config.scs)
    beta = float(beta_dmrs)
    alpha = float(getattr(config, "CNNSmoothingAlpha", 0.0) or 0.0)
    # extension (Smoothing="mmse"): delay spread defaults to the normal cyclic prefix, noise-to-signal to -20 dB
    mmse_tau = getattr(config, "MMSEDelaySpread", None)
    mmse_tau = (float(cp[1]) * 1e-3 if cp.size > 1 else 0.0) if mmse_tau is None else float(mmse_tau)
    mmse_nsr = getattr(config, "MMSENoiseToSignal", None)
    mmse_nsr = 0.01 if mmse_nsr is None else float(mmse_nsr)
    n_cdm = (n_layers + 1) // 2

    d1, r1, m1 = _hop_fields(hop1, n_cdm)
    d2, r2, m2 = _hop_fields(hop2, n_cdm)
    has_hop2 = d2.size != 0 and int(d2.sum()) != 0
    hops = [(hop1, d1, r1, m1)]
    if has_hop2:
        assert not bool(np.any(d1 & d2)), "Hops should not overlap."
        assert r1.shape == r2.shape and bool(np.all(r1 == r2)), "The DM-RS mask should be the same for the two hops."
        hops.append((hop2, d2, r2, m2))
    geo = tuple((d, r, m, int(h.PRBstart), int(h.nPRBs), int(h.startSymbol), int(h.nAllocatedSymbols)) for h, d, r, m in hops)
    key = (n_layers, n_prb_grid, n_sym, smoothing, cfo_comp, interp, scs, beta, alpha, mmse_tau, mmse_nsr, cp14.tobytes(),
           tuple((d.tobytes(), r.tobytes(), r.shape, m.tobytes(), ps, npr, ss, na) for d, r, m, ps, npr, ss, na in geo))
    vals = dict(n_layers=n_layers, n_prb_grid=n_prb_grid, n_sym=n_sym, smoothing=smoothing, cfo_comp=cfo_comp, interp=interp,
                scs=scs, beta=beta, alpha=alpha, mmse_tau=mmse_tau, mmse_nsr=mmse_nsr, cp14=cp14, geo=geo, n_cdm=n_cdm)
    return vals, key


def _fill_desc(v, device_index: int):
    """``ce_plan_desc`` from ``_resolve``'s values.  Returns ``(desc, keepalive)``."""
    n_sym, n_prb_grid, n_cdm = v["n_sym"], v["n_prb_grid"], v["n_cdm"]
    desc = _lib.PlanDesc()
    desc.abi_version = _lib.CE_ABI_VERSION
    desc.device = device_index
    desc.n_prb_grid, desc.n_sym, desc.n_layers, desc.n_hops = n_prb_grid, n_sym, v["n_layers"], len(v["geo"])
    desc.smoothing, desc.cfo_compensate, desc.interp = _lib.SMOOTHING[v["smoothing"]], int(v["cfo_comp"]), _lib.INTERP[v["interp"]]
    desc.scs_hz, desc.beta_dmrs, desc.cnn_smoothing_alpha = v["scs"], v["beta"], v["alpha"]
    desc.mmse_delay_spread_s, desc.mmse_noise_to_signal = v["mmse_tau"], v["mmse_nsr"]
    for i in range(14):
        desc.cp_ms[i] = v["cp14"][i]
    keep = []
    for hi, (d, r, m, prb_start, n_prbs, start_symbol, n_alloc) in enumerate(v["geo"]):
        hd = desc.hop[hi]
        if d.size != n_sym:
            raise ValueError(f"hop {hi + 1}: DMRSsymbols has {d.size} entries, grid has {n_sym} symbols")
        if r.ndim != 2 or r.shape[0] != 12 or r.shape[1] < n_cdm:
            raise ValueError(f"hop {hi + 1}: DMRSREmask must be (12, >= {n_cdm}), got {r.shape}")
        if m.size != n_prb_grid:
            raise ValueError(f"hop {hi + 1}: maskPRBs has {m.size} entries, grid has {n_prb_grid} PRBs")
        for s in range(n_sym):
            hd.dmrs_symbols[s] = int(d[s])
        for c in range(n_cdm):
            hd.re_mask[c] = int(sum(1 << rr for rr in range(12) if r[rr, c]))
        buf = (C.c_uint8 * n_prb_grid).from_buffer_copy(np.ascontiguousarray(m, np.uint8).tobytes())
        keep.append(buf)
        hd.mask_prbs = C.cast(buf, C.POINTER(C.c_uint8))
        hd.prb_start, hd.n_prbs = prb_start, n_prbs
        hd.start_symbol, hd.n_alloc_symbols = start_symbol, n_alloc
    return desc, keep


def build_desc(hop1, hop2, config, beta_dmrs, n_layers: int, n_prb_grid: int, n_sym: int, device_index: int = 0,
               interp: str = "linear"):
    """Validate like the reference and fill a ``ce_plan_desc``.  Returns ``(desc, key, keepalive)``; needs no GPU."""
    vals, key = _resolve(hop1, hop2, config, beta_dmrs, n_layers, n_prb_grid, n_sym, interp)
    desc, keep = _fill_desc(vals, device_index)
    return desc, (device_index,) + key, keep


def derive_host(hop1, hop2, config, beta_dmrs, n_layers: int, n_prb_grid: int, n_sym: int, interp: str = "linear"):
    """Host-only view of what a plan would contain (``ce_plan_derive_host``): works without a GPU."""
    lib = _lib.load()
    desc, _, keep = build_desc(hop1, hop2, config, beta_dmrs, n_layers, n_prb_grid, n_sym, 0, interp)
    view = _lib.PlanHostView()
    rc = lib.ce_plan_derive_host(C.byref(desc), C.byref(view))
    del keep
    if rc != 0:
        _raise_for(rc)
    return view


def make_plan(hop1, hop2, config, beta_dmrs, n_layers: int, n_prb_grid: int, n_sym: int,
              device: torch.device | str | int | None = None, interp: str = "linear") -> Plan:
    """Resolve (hop1, hop2, config) into a cached GPU plan (exceptions: see ``build_desc``)."""
    lib = _lib.load()
    # validation first: the reference's error cases must surface even where no GPU is visible
    vals, key = _resolve(hop1, hop2, config, beta_dmrs, n_layers, n_prb_grid, n_sym, interp)
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the estimator runs on a ROCm GPU only (no CPU fallback)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index,) + key
    with _PLAN_CACHE_LOCK:                      # host threads may share the cache (a get / move_to_end pair must not straddle another thread's eviction)
        plan = _PLAN_CACHE.get(key)
        if plan is not None:                    # steady state: no descriptor, no ctypes buffers, no library call
            _PLAN_CACHE.move_to_end(key)
            return plan
    desc, keep = _fill_desc(vals, device.index)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = lib.ce_plan_create(C.byref(desc), C.byref(handle))
    del keep
    if rc != 0:
        _raise_for(rc)
    info = _lib.PlanInfo()
    lib.ce_plan_get_info(handle, C.byref(info))
    plan = Plan(handle.value, info, key, device, n_layers, n_sym)
    with _PLAN_CACHE_LOCK:
        _PLAN_CACHE[key] = plan                 # (two threads that missed together both built the plan: the later one stays, the other lives as long as its caller holds it)
        while len(_PLAN_CACHE) > _PLAN_CACHE_MAX:
            _PLAN_CACHE.popitem(last=False)
    return plan


def _strides4(t: torch.Tensor):
    return (C.c_int64 * 4)(*[int(s) for s in t.stride()])


def _batch_args(plan: Plan, received_rg: torch.Tensor, pilots: torch.Tensor):
    if received_rg.dim() != 4:
        raise ValueError("received_rg must be [slots, ports, n_sc, n_sym]")
    B, R, n_sc, n_sym = received_rg.shape
    if n_sc != plan.n_sc or n_sym != plan.n_sym:
        raise ValueError(f"grid is {n_sc}x{n_sym}, plan expects {plan.n_sc}x{plan.n_sym}")
    if received_rg.device != plan.device or pilots.device != plan.device:
        raise ValueError(f"tensors must live on {plan.device}")
    if received_rg.dtype != torch.complex64 or pilots.dtype != torch.complex64:
        raise ValueError("received_rg and pilots must be complex64")
    if pilots.dim() == 3:
        pilots = pilots.unsqueeze(0).expand(B, -1, -1, -1)
    if pilots.dim() != 4 or pilots.shape[0] != B or tuple(pilots.shape[1:]) != (plan.n_re, plan.n_dmrs_total, plan.n_layers):
        raise ValueError(f"pilots must be [{B}|-, {plan.n_re}, {plan.n_dmrs_total}, {plan.n_layers}], got {tuple(pilots.shape)}")
    if any(s < 0 for s in received_rg.stride()) or any(s < 0 for s in pilots.stride()):
        raise ValueError("negative strides are not supported")
    return B, R, pilots


def estimate_with_plan(plan: Plan, received_rg: torch.Tensor, pilots: torch.Tensor, out: Optional[tuple] = None):
    """One fused launch on the current stream of ``plan.device``.  Returns
    ``(ch_est[B,R,n_sc,n_sym,L] complex64, noise[B,R], rsrp[B,R], epre[B,R], ta[B,R], cfo_hz[B,R])``
    (float64; ``cfo_hz`` is NaN-filled when ``plan.cfo_estimated`` is False).  ``out`` may pass the
    same six tensors to be overwritten (no allocation in steady state)."""
    lib = _lib.load()
    B, R, pilots4 = _batch_args(plan, received_rg, pilots)
    dev = plan.device
    if out is None:
        ch = torch.empty((B, R, plan.n_sc, plan.n_sym, plan.n_layers), dtype=torch.complex64, device=dev)
        sc = torch.empty((5, B, R), dtype=torch.float64, device=dev)
        out = (ch, sc[0], sc[1], sc[2], sc[3], sc[4])
    ch = out[0]
    if not ch.is_contiguous() or ch.dtype != torch.complex64 or tuple(ch.shape) != (B, R, plan.n_sc, plan.n_sym, plan.n_layers):
        raise ValueError("out[0] must be a contiguous complex64 [B,R,n_sc,n_sym,L] tensor")
    for t in out[1:]:
        if not t.is_contiguous() or t.dtype != torch.float64 or t.numel() != B * R:
            raise ValueError("scalar outputs must be contiguous float64 [B,R] tensors")
    if B * R == 0:
        return out                                           # empty batch: nothing to launch
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.ce_estimate_batch(plan._handle, received_rg.data_ptr(), _strides4(received_rg), pilots4.data_ptr(),
                               _strides4(pilots4), B, R, ch.data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                               out[3].data_ptr(), out[4].data_ptr(), out[5].data_ptr(), stream)
    if rc != 0:
        _raise_for(rc)
    return out


def estimate_stages(plan: Plan, received_rg: torch.Tensor, pilots: torch.Tensor, n_hops: int):
    """Diagnostic launch (``ce_estimate_batch_stages``): the ordinary six outputs plus
    ``stage_estimates[B,R,2,n_hops,L,n_re]`` (pilot-RE channel estimate after LS / de-spread and after smoothing) and
    ``stage_scalars[B,R,n_hops,2]`` (hop CFO normalised to the SCS, NaN where not estimated; TA arg-max bin)."""
    lib = _lib.load()
    B, R, pilots4 = _batch_args(plan, received_rg, pilots)
    dev = plan.device
    ch = torch.empty((B, R, plan.n_sc, plan.n_sym, plan.n_layers), dtype=torch.complex64, device=dev)
    sc = torch.empty((5, B, R), dtype=torch.float64, device=dev)
    st_p = torch.full((B, R, 2, n_hops, plan.n_layers, plan.n_re), float("nan"), dtype=torch.complex64, device=dev)
    st_s = torch.full((B, R, n_hops, 2), float("nan"), dtype=torch.float64, device=dev)
    if B * R:
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.ce_estimate_batch_stages(plan._handle, received_rg.data_ptr(), _strides4(received_rg), pilots4.data_ptr(),
                                          _strides4(pilots4), B, R, ch.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(),
                                          sc[2].data_ptr(), sc[3].data_ptr(), sc[4].data_ptr(), st_p.data_ptr(), st_s.data_ptr(), stream)
        if rc != 0:
            _raise_for(rc)
    return (ch, sc[0], sc[1], sc[2], sc[3], sc[4]), st_p, st_s


def time_with_plan(plan: Plan, received_rg: torch.Tensor, pilots: torch.Tensor, out: tuple, warmup: int, iters: int) -> float:
    """Average launch duration in ms, HIP events on the current stream (bench.py's roofline leg)."""
    lib = _lib.load()
    B, R, pilots4 = _batch_args(plan, received_rg, pilots)
    ms = C.c_double(0.0)
    stream = torch.cuda.current_stream(plan.device).cuda_stream
    rc = lib.ce_time_batch(plan._handle, received_rg.data_ptr(), _strides4(received_rg), pilots4.data_ptr(),
                           _strides4(pilots4), B, R, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                           out[3].data_ptr(), out[4].data_ptr(), out[5].data_ptr(), stream, warmup, iters, C.byref(ms))
    if rc != 0:
        _raise_for(rc)
    return ms.value


def estimate(received_rg: torch.Tensor, pilots: torch.Tensor, beta_dmrs, hop1, hop2, config, *,
             interp: str = "linear", out: Optional[tuple] = None):
    """Batched estimator: ``received_rg`` is logically ``[slots, ports, n_sc, n_sym]`` with arbitrary
    strides (a ``[slots, ports, n_sym, n_sc]`` buffer viewed through ``.permute(0,1,3,2)`` gives
    coalesced pilot loads); ``pilots`` is ``[n_re, n_dmrs, L]`` (shared) or ``[slots, n_re, n_dmrs, L]``.
    Returns the reference's 6-tuple with leading ``[slots, ports]`` axes; ``cfo_hz`` is an empty
    float64 tensor when no hop has two DM-RS symbols (T:931-933)."""
    if not torch.is_complex(received_rg):
        received_rg = received_rg.to(torch.complex64)
    if not torch.is_complex(pilots):
        pilots = pilots.to(torch.complex64)
    received_rg = received_rg.to(torch.complex64)
    pilots = pilots.to(device=received_rg.device, dtype=torch.complex64)
    if received_rg.dim() != 4:
        raise ValueError("received_rg must be [slots, ports, n_sc, n_sym]")
    n_sc, n_sym = received_rg.shape[2], received_rg.shape[3]
    if n_sc % 12:
        raise ValueError("n_sc must be a multiple of 12")
    plan = make_plan(hop1, hop2, config, beta_dmrs, pilots.shape[-1], n_sc // 12, n_sym, received_rg.device, interp)
    res = estimate_with_plan(plan, received_rg, pilots, out)
    # EXTENSION (no reference counterpart): `config.Denoiser`, a srsran_ce_pytorch_amd.denoiser.Denoiser, post-processes
    # the channel estimate in place on the same stream -- the reference reads optional attributes the same way (C:864)
    denoiser = getattr(config, "Denoiser", None)
    if denoiser is not None:
        denoiser(res[0])
    if not plan.cfo_estimated:
        return res[:5] + (torch.empty((0,), dtype=torch.float64, device=received_rg.device),)
    return res


def srs_channel_estimator(received_rg, pilots, beta_dmrs, hop1, hop2, config, *, interp: str = "linear"
                          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Drop-in for the reference's single-slot, single-port entry point (T:745-772): same
    arguments, same 6-tuple (grid in the input's complex dtype and device, 0-d float64 scalars,
    shape-(0,) ``cfo`` when not estimated).  The arithmetic runs on the current ROCm device.
    ``interp="cnn"`` is ``src/ce_dl_cnn.py``'s entry point of the same signature (C:802)."""
    received_rg = torch.as_tensor(received_rg)
    pilots = torch.as_tensor(pilots)
    if not torch.is_complex(received_rg):
        received_rg = received_rg.to(torch.complex64)          # T:776-779
    if received_rg.dim() != 2 or pilots.dim() != 3:
        raise ValueError("received_rg must be (n_sc, n_sym) and pilots (n_re, n_dmrs, n_layers)")
    in_dev, in_dtype = received_rg.device, received_rg.dtype
    if in_dtype == torch.complex128:
        # A deliberate, permanent narrowing: the reference keeps a complex128 grid's dtype (T:773-779) but
        # estimates in the PILOTS' dtype (T:556-558) -- complex64 for its own harness (validate_case0.py:40-47) -- and
        # interpolates in float64.  The HIP path estimates in complex64 and casts back: <= 3e-7 on the grid, <= 2e-6 on the
        # scalars, TA identical, against the real reference's complex128 outputs for both pilot dtypes (tests/golden/c128*.npz).
        # Said out loud rather than done silently.
        warnings.warn("srs_channel_estimator: complex128 grid is estimated in complex64 on the GPU and cast back "
                      "(pinned <= 3e-7 from the reference's complex128 outputs; see INTEGRATION.md)", RuntimeWarning, stacklevel=2)
    dev = in_dev if in_dev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    rg = received_rg.to(device=dev, dtype=torch.complex64)[None, None]
    res = estimate(rg, pilots.to(dev), beta_dmrs, hop1, hop2, config, interp=interp)
    ch = res[0][0, 0].to(device=in_dev, dtype=in_dtype)
    scal = [t.reshape(()).to(in_dev) if t.numel() else t.to(in_dev) for t in res[1:]]
    return (ch, *scal)
