"""Slot-batch sharding across the GPUs of a node: slots are independent, so each rank
owns a contiguous range of the slot axis (all Rx ports of a slot stay together) and the data path needs no
collective.  `torch.distributed` is used for the bench's barrier / max-over-ranks timing only."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_slots(n_slots: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the slot axis owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_slots, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(values, device="cpu"):
    """Elementwise MAX of a list of floats over all ranks (measurement only, never data path)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():    # also at world size 1 (bench.py's CE_BENCH_FORCE_PG rehearsal of the RCCL path)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def aggregate_slots_per_second(slots_per_rank: int, steps: int, elapsed_s: float, world: int) -> float:
    """Whole-job throughput: slots all ranks processed / max-over-ranks wall time."""
    return world * slots_per_rank * steps / elapsed_s


# ------------------------------------------------------------------------------------------------------------------
# One process, several devices ("one host thread + one stream per GPU; plan object replicated per
# device").  The multi-process form (one rank per GPU under torch.distributed.run) is bench.py's; this is the library
# entry point for a caller that owns all GPUs of a node from one Python process.
# ------------------------------------------------------------------------------------------------------------------
_SHARD_STREAMS = {}


def _shard_stream(device: torch.device, k: int) -> "torch.cuda.Stream":
    """The k-th launch stream of `device` (created once; two shards on one device get two streams)."""
    key = (device.index, k)
    s = _SHARD_STREAMS.get(key)
    if s is None:
        s = _SHARD_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def split_slots(received_rg: torch.Tensor, pilots: torch.Tensor, devices):
    """Contiguous slot ranges of a `[slots, ports, n_sc, n_sym]` batch (and its `[slots, n_re, n_dmrs, L]` or shared
    `[n_re, n_dmrs, L]` pilots) copied to `devices` (`shard_slots` arithmetic: sizes differ by at most one; all Rx ports
    of a slot stay together).  Copies are asynchronous on each destination's current stream.  Returns
    `(rx_shards, pilots_shards)` for `estimate_sharded`; strides of the source are kept (a `[.., sym, sc]` buffer viewed
    through `.permute(0, 1, 3, 2)` stays subcarrier-contiguous on the destination)."""
    devs = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    n = received_rg.shape[0]
    rx_s, pil_s = [], []
    for r, d in enumerate(devs):
        a, b = shard_slots(n, r, len(devs))
        part = received_rg[a:b]
        dst = torch.empty_strided(part.shape, part.stride(), dtype=part.dtype, device=d) if part.numel() else torch.empty(part.shape, dtype=part.dtype, device=d)
        dst.copy_(part, non_blocking=True)
        rx_s.append(dst)
        pil_s.append((pilots if pilots.dim() == 3 else pilots[a:b]).to(d, non_blocking=True))
    return rx_s, pil_s


def estimate_sharded(rx_shards, pilots_shards, beta_dmrs, hop1, hop2, config, devices=None, *, interp: str = "linear", outs=None):
    """Batched estimation over several devices from ONE host thread: shard i (`rx_shards[i]`: `[slots_i, ports, n_sc,
    n_sym]`, `pilots_shards[i]`, both resident on `devices[i]`; default: each shard's own device) gets the plan of its
    device (`estimator.make_plan`, cached per device) and its own launch stream; the launches are issued round-robin
    with NO synchronisation between devices and no collective -- slots are independent (src/ce_rule_tensorized.py:745-937
    keeps no cross-call state).  A device may appear more than once (two shards, two streams on it).

    Each launch stream first waits (on the GPU) for the device's current stream, so inputs produced there are complete,
    and the device's current stream afterwards waits for the launch stream, so the results can be consumed there
    without a host synchronisation; the host never blocks.  Returns one result tuple per shard -- the six outputs of
    `estimator.estimate_with_plan` (`ch_est[slots_i, ports, n_sc, n_sym, L]`, `noise, rsrp, epre, ta, cfo_hz` as
    `[slots_i, ports]`; `cfo_hz` NaN-filled when not estimated) -- left on the shard's device.  `outs[i]` may pass
    the six tensors of shard i to be overwritten."""
    from . import estimator as E

    n = len(rx_shards)
    if len(pilots_shards) != n or (outs is not None and len(outs) != n):
        raise ValueError("rx_shards, pilots_shards (and outs) must have one entry per shard")
    devs = [t.device for t in rx_shards] if devices is None else \
        [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    if len(devs) != n:
        raise ValueError(f"{n} shards but {len(devs)} devices")
    seen, plans, streams = {}, [], []
    for i, (rx, pil, d) in enumerate(zip(rx_shards, pilots_shards, devs)):
        if d.type != "cuda":
            raise RuntimeError("the estimator runs on ROCm GPUs only (no CPU fallback)")
        if d.index is None:
            d = devs[i] = torch.device("cuda", torch.cuda.current_device())
        if rx.device != d or pil.device != d:
            raise ValueError(f"shard {i} lives on {rx.device} / {pil.device}, expected {d}")
        if rx.dim() != 4 or rx.shape[2] % 12:
            raise ValueError("every shard must be [slots, ports, n_sc, n_sym] with n_sc a multiple of 12")
        plans.append(E.make_plan(hop1, hop2, config, beta_dmrs, pil.shape[-1], rx.shape[2] // 12, rx.shape[3], d, interp))
        k = seen.get(d.index, 0)
        seen[d.index] = k + 1
        streams.append(_shard_stream(d, k))
    results = []
    for i in range(n):                       # round-robin issue: one launch per shard, nothing waits on the host
        d, st = devs[i], streams[i]
        cur = torch.cuda.current_stream(d)
        with torch.cuda.device(d):
            out = outs[i] if outs is not None else None
            if out is None:                  # allocated on the device's current stream, which later waits for the launch stream
                ch = torch.empty((rx_shards[i].shape[0], rx_shards[i].shape[1], plans[i].n_sc, plans[i].n_sym, plans[i].n_layers),
                                 dtype=torch.complex64, device=d)
                sc = torch.empty((5,) + tuple(rx_shards[i].shape[:2]), dtype=torch.float64, device=d)
                out = (ch, sc[0], sc[1], sc[2], sc[3], sc[4])
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                res = E.estimate_with_plan(plans[i], rx_shards[i], pilots_shards[i], out)
            for t in (rx_shards[i], pilots_shards[i]) + tuple(res):
                t.record_stream(st)
        results.append(res)
    for i in range(n):
        torch.cuda.current_stream(devs[i]).wait_stream(streams[i])
    return results
