#!/usr/bin/env python3
"""Headline benchmark: slots/sec of the fused PUSCH DM-RS channel-estimation path at 273 PRB.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 under torch.distributed.run, one
rank per GPU).  A *step* is one pass of the hot path over one resident batch of synthetic slots
(one fused kernel launch per rank).  The slot batch is sharded across ranks with NO data-path
collective (slots are independent) -> weak scaling: every rank owns ``--slots`` slots.

Workloads (BASELINE.json configs):
  pusch273_4rx_filter   configs[2]: 273 PRB / 4 Rx / 8192 slots per GPU, Smoothing="filter" -- the
                        reference's own frequency smoothing (it has no MMSE/Wiener mode).
                        This is the configuration the metric ("273-PRB PUSCH, 4 Rx") is quoted on.
  pusch273_1rx_none     configs[1]: LS + linear interpolation, 273 PRB / 1 Rx / 1024 slots.
  pusch273_4rx_mmse     EXTENSION (parity unpinned): block LMMSE/Wiener smoothing on f32 MFMA instead of the RC FIR.
  pusch273_4rx_denoise  EXTENSION (parity unpinned): estimation + the fp16 Conv2d denoiser of include/ce_denoise.h on MFMA
                        (random weights); its roofline object is MFMA-bound and describes the denoiser kernel.
  pusch273_4rx_cnn      configs[4]: the ce_dl_cnn.py variant (fixed-weight 1-D in-painting instead of linear
                        interpolation; the reference has no Conv2d / fp16 / learned weights).

Prints ONE JSON line on rank 0 with the driver's fields plus ``roofline`` (HBM; algorithmic bytes
per launch / HIP-event launch time) and, at N=1, ``cpu_baseline`` (CPU ports of the reference's algorithm --
the loop-style ce_rule_baseline form and the tensorized form -- timed on the host cores on bounded samples of the same
workload) and ``secondary`` (the other parity-pinned workloads, a few launches each, followed by the two extensions
without a reference counterpart, each marked ``"parity": "unpinned extension"``).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL bootstrap at N > 1)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "pusch273_4rx_filter": dict(smoothing="filter", ports=4, slots=8192),
    "pusch273_1rx_none": dict(smoothing="none", ports=1, slots=1024),
    # configs[4] as the reference actually implements it: ce_dl_cnn's fixed 3-tap in-painting
    "pusch273_4rx_cnn": dict(smoothing="filter", ports=4, slots=8192, interp="cnn"),
    # configs[2] as BASELINE.json words it ("MMSE Wiener filter on"): an EXTENSION -- the reference has no such mode
    # (SURVEY 0.4), its only oracle is the build's own numpy restatement => parity unpinned, reported separately
    "pusch273_4rx_mmse": dict(smoothing="mmse", ports=4, slots=8192),
    # configs[4] as BASELINE.json words it ("Conv2d over RE grid, fp16, MFMA path"): an EXTENSION with random weights --
    # the reference has no learned denoiser; estimation (filter) + the 3-layer Conv2d post-processor per step
    "pusch273_4rx_denoise": dict(smoothing="filter", ports=4, slots=8192, denoise=True),
}
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 MFMA (same guide; 2:1-sparsity figures are never the yardstick)
DENOISE_FLOP_PER_PIXEL = 2 * 9 * (2 * 16 + 16 * 16 + 16 * 2)   # useful MACs x 2 of the three 3x3 layers


def _cpu_worker(args):
    """One CPU restatement of the reference algorithm on one slot's ports, repeated; returns (items done, seconds).
    flavour: "tensorized" = oracle/ce_oracle.py (ce_rule_tensorized.py), "baseline_loop" = oracle/ce_oracle_baseline.py
    (the loop-style ce_rule_baseline.py BASELINE.json names), "cnn" = the ce_dl_cnn.py fill, "+denoise" = extension."""
    case, n_ports, reps, seed, flavour, denoise = args
    sys.path.insert(0, str(ROOT / "oracle"))
    import ce_oracle as O
    import ce_oracle_baseline as OB
    from srsran_ce_pytorch_amd import synth as S

    b = S.build_case(dict(case, seed=seed), n_ports)
    if denoise:                                                         # extension workload: its own numpy restatement
        import ce_denoise_oracle as DO
        from srsran_ce_pytorch_amd.denoiser import random_weights
        weights = random_weights(0)

    def one(r):
        if flavour == "baseline_loop":
            out = OB.srs_channel_estimator(b.grids[r], b.pilots, b.beta, b.hop1, b.hop2, b.config)
        else:
            out = O.srs_channel_estimator(b.grids[r], b.pilots, b.beta, b.hop1, b.hop2, b.config,
                                          interp="cnn" if flavour == "cnn" else "linear")
        if denoise:
            DO.denoise(out[0], weights)

    for _ in range(2):                                                  # untimed warm-up (page faults, caches)
        for r in range(n_ports):
            one(r)
    t0 = time.perf_counter()
    for _ in range(reps):
        for r in range(n_ports):
            one(r)
    return reps * n_ports, time.perf_counter() - t0


def _cpu_leg(case, n_ports, flavour, denoise, target_core_seconds, cores):
    import multiprocessing as mp

    n1, t1 = _cpu_worker((case, n_ports, 2, 999, flavour, denoise))     # per-item estimate on one core
    per_item = t1 / n1
    reps = max(1, int(target_core_seconds / cores / (per_item * n_ports)))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(case, n_ports, reps, 1000 + i, flavour, denoise) for i in range(cores)])
    wall = time.perf_counter() - t0
    busy = max(r[1] for r in res)
    slots = sum(r[0] for r in res) / n_ports
    return dict(slots_per_s=slots / busy, slots=int(slots), busy_s=busy, wall_s=wall, ms_per_slot_port_one_core=per_item * 1e3)


def cpu_baseline(case, n_ports, target_core_seconds=20.0, interp="linear", denoise=False):
    """Bounded samples of the same workload on the host cores (forks happen BEFORE any GPU init): the loop-style
    restatement of ce_rule_baseline.py -- the baseline north_star names, `value` -- and the tensorized restatement next
    to it, plus BASELINE.json configs[0] (25 PRB / 1 DM-RS / 1 Rx / LS only, one slot at a time on one core)."""
    from srsran_ce_pytorch_amd import synth as S

    cores = min(16, len(os.sched_getaffinity(0)))
    fast = _cpu_leg(case, n_ports, "cnn" if interp == "cnn" else "tensorized", denoise, target_core_seconds * 0.4, cores)
    out = dict(unit="slots/s", cores=cores, kind="port")
    if interp == "cnn" or denoise:                                      # these workloads have no loop-style reference form
        name = "ce_dl_cnn" if interp == "cnn" else "ce_rule_tensorized"
        out.update(value=fast["slots_per_s"], flavour=name + (" + Conv2d denoiser (extension)" if denoise else ""))
        loop = None
    else:
        loop = _cpu_leg(case, n_ports, "baseline_loop", False, target_core_seconds * 0.6, cores)
        out.update(value=loop["slots_per_s"], flavour="ce_rule_baseline (loop-style)", tensorized_value=fast["slots_per_s"],
                   baseline_loop_ms_per_slot_port=loop["ms_per_slot_port_one_core"],
                   tensorized_ms_per_slot_port=fast["ms_per_slot_port_one_core"])
    c0 = S.config1_case()                                               # configs[0], the reference's own CPU-runnable case
    n0, t0 = _cpu_worker((c0, 1, 20, 7, "baseline_loop", False))
    out["config0_ms"] = t0 / n0 * 1e3
    legs = [("oracle/ce_oracle_baseline.py (numpy port of ce_rule_baseline, per-gap / per-layer loops)", loop),
            (f"oracle/ce_oracle.py (numpy port of {'ce_dl_cnn' if interp == 'cnn' else 'ce_rule_tensorized'}"
             f"{' + oracle/ce_denoise_oracle.py' if denoise else ''})", fast)]
    out["sample"] = "; ".join(f"{l['slots']} slots x {n_ports} ports of the same 273-PRB workload through {what}, {cores} worker "
                              f"processes, {l['busy_s']:.1f} s busy / {l['wall_s']:.1f} s wall, {l['ms_per_slot_port_one_core']:.2f} ms per slot-port on one core"
                              for what, l in legs if l is not None) + \
        f"; config0_ms = configs[0] (25 PRB in a 52-PRB grid, 1 DM-RS symbol, 1 Rx, LS only) through the loop-style port, one core"
    return out


def kernel_source_sha():
    """Hash of the sources the estimation kernels are built from (same recipe as tools/distill_round.py)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("ce_estimate_kernel.h", "ce_plan.h", "ce_api.hip", "ce_inst.inc"):
        h.update((ROOT / "srsran_ce_pytorch_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def time_steps(step, barrier, steps, warmup, make_events=None):
    """The driver's timing contract for one rank: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    `barrier()` (process-group barrier + device synchronise) on both sides.  Returns (wall seconds of the bracket,
    average ms per step between two events recorded on the launch stream, or None without `make_events`)."""
    for _ in range(warmup):
        step()
    barrier()
    ev = make_events() if make_events is not None else None
    t0 = time.perf_counter()
    if ev is not None:
        ev[0].record()                                              # same stream the kernel is launched on
    for _ in range(steps):
        step()
    if ev is not None:
        ev[1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, (ev[0].elapsed_time(ev[1]) / steps if ev is not None else None)


def secondary_lines(E, S, plan, case, rx, pilots, out, n_slots, n_ports, dev, iters=5):
    """The other parity-pinned workloads of BASELINE.json, a few launches each after the headline (N=1 only), so the
    driver's record carries them: configs[1] (LS + linear interpolation, 1024 slots x 1 Rx), configs[4] as the reference
    implements it (ce_dl_cnn in-painting), and the headline fed in the reference's own [sc][sym] grid layout (what the
    drop-in shim receives; its algorithmic bytes are the same, its HBM traffic is the whole grid)."""
    import torch

    def timed(pl, rx_, pil_, out_, slots, ports, name, layout, workload_case):
        ms = min(E.time_with_plan(pl, rx_, pil_, out_, 1, iters) for _ in range(2))
        if ms < 1.0:   # a sub-millisecond launch: five of them are over before the clocks have settled after the set-up above
            E.time_with_plan(pl, rx_, pil_, out_, 0, int(30.0 / ms))
            ms = min(E.time_with_plan(pl, rx_, pil_, out_, 0, 20) for _ in range(3))
        b = slots * (ports * pl.alg_bytes_per_item + pl.pilot_bytes_per_slot)
        ach = b / (ms * 1e-3) / 1e9
        return {"workload": name, "slots": slots, "rx_ports": ports, "smoothing": workload_case["smoothing"], "ms_per_step": ms,
                "slots_per_s": slots / (ms * 1e-3), "rx_layout": layout,
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "alg_bytes_per_launch": b}}

    res = []
    # configs[1]: its own small resident batch
    w1 = WORKLOADS["pusch273_1rx_none"]
    c1 = S.bench_case(w1["smoothing"], 1, seed=4321)
    h1, h2, cfg1 = S.numpy_hops(c1)
    p1 = E.make_plan(h1, h2, cfg1, c1["beta"], 1, c1["n_prb_grid"], c1["n_sym"], dev)
    rx1, pil1 = S.torch_inputs(c1, w1["slots"], w1["ports"], dev, seed=4321)
    out1 = E.estimate_with_plan(p1, rx1, pil1)
    res.append(timed(p1, rx1, pil1, out1, w1["slots"], w1["ports"], "pusch273_1rx_none", "[slot][port][sym][sc]", c1))
    del rx1, pil1, out1
    # configs[4] (reference form): same inputs and outputs as the headline, in-painting instead of linear interpolation
    hh1, hh2, cfgh = S.numpy_hops(case)
    pc = E.make_plan(hh1, hh2, cfgh, case["beta"], 1, case["n_prb_grid"], case["n_sym"], dev, "cnn")
    res.append(timed(pc, rx, pilots, out, n_slots, n_ports, "pusch273_4rx_cnn", "[slot][port][sym][sc]", case))
    # headline in the reference layout: dense [slot][port][sc][sym] grids
    rx_ref = rx.contiguous()
    res.append(timed(plan, rx_ref, pilots, out, n_slots, n_ports, "pusch273_4rx_filter", "[slot][port][sc][sym] (reference layout)", case))
    del rx_ref
    # The two EXTENSIONS north_star names and the reference does not have -- parity unpinned, never part of the
    # headline: block LMMSE smoothing in place of the RC FIR (HBM-bound like the rest), and the fp16 Conv2d denoiser kernel
    # alone on the resident batch of estimates (MFMA-bound; random weights).
    pm = E.make_plan(hh1, hh2, _with_smoothing(cfgh, "mmse"), case["beta"], 1, case["n_prb_grid"], case["n_sym"], dev)
    e = timed(pm, rx, pilots, out, n_slots, n_ports, "pusch273_4rx_mmse", "[slot][port][sym][sc]", dict(case, smoothing="mmse"))
    e["parity"] = "unpinned extension"
    res.append(e)
    from srsran_ce_pytorch_amd.denoiser import Denoiser, random_weights
    dn = Denoiser(random_weights(0), dev)
    dn(out[0])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(2):
        ev0.record()
        for _ in range(3):
            dn(out[0])
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / 3)
    flops = float(n_slots * n_ports * case["n_prb_grid"] * 12 * case["n_sym"] * DENOISE_FLOP_PER_PIXEL)
    res.append({"workload": "conv2d_denoiser_kernel", "slots": n_slots, "rx_ports": n_ports, "ms_per_step": best,
                "slots_per_s": n_slots / (best * 1e-3), "parity": "unpinned extension",
                "roofline": {"bound": "mfma", "achieved": flops / (best * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops / (best * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, "alg_flop_per_launch": flops}})
    torch.cuda.synchronize()
    # The reference harness's own shapes (52-PRB grids, 3-PRB allocations: scripts/validation/validate_case{0,4,8}.py; case 4 describes
    # BOTH hops over the whole slot) and the narrow multi-layer two-hop shape: small resident batches of their own, the plans'
    # own choice of kernel (the wave-per-item kernel, csrc/ce_narrow_kernel.h)
    H, CS = S.hop_spec, S.case_spec
    for nm, cs in (("harness52_case4like_2hops_3prb_4rx", CS("hc4", 52, [H([0, 4], 3, 3, 0, 14), H([8, 12], 28, 3, 0, 14)], scs=15e3)),
                   ("harness52_case0like_3prb_4dmrs_4rx", CS("hc0", 52, [H([0, 4, 8, 12], 40, 3)], scs=15e3)),
                   ("narrow52_2layers_2hops_12prb_4rx", CS("h2l2n", 52, [H([1, 5], 3, 12, 0, 7), H([8, 12], 30, 12, 7, 7)], n_layers=2))):
        g1, g2, gcfg = S.numpy_hops(cs)
        pn = E.make_plan(g1, g2, gcfg, cs["beta"], cs["n_layers"], cs["n_prb_grid"], cs["n_sym"], dev)
        rxn, piln = S.torch_inputs(cs, n_slots, n_ports, dev, seed=77)
        outn = E.estimate_with_plan(pn, rxn, piln)
        e = timed(pn, rxn, piln, outn, n_slots, n_ports, nm, "[slot][port][sym][sc]", cs)
        e["n_prb_grid"], e["layers"], e["hops"] = cs["n_prb_grid"], cs["n_layers"], len(cs["hops"])
        res.append(e)
        del rxn, piln, outn
    torch.cuda.synchronize()
    return res


def _with_smoothing(cfg, smoothing):
    import copy

    c = copy.copy(cfg)
    c.Smoothing = smoothing
    return c


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pusch273_4rx_filter", choices=sorted(WORKLOADS))
    ap.add_argument("--slots", type=int, default=None, help="slots per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads timed after the headline (N=1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    from srsran_ce_pytorch_amd import synth as S

    wl = WORKLOADS[args.workload]
    n_slots = args.slots or wl["slots"]
    n_ports = wl["ports"]
    case = S.bench_case(wl["smoothing"], 1, seed=1234 + rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(case, n_ports, interp=wl.get("interp", "linear"), denoise=bool(wl.get("denoise")))   # before any GPU initialisation (forks)

    import torch
    import torch.distributed as dist
    from srsran_ce_pytorch_amd import estimator as E
    from srsran_ce_pytorch_amd.sharding import aggregate_slots_per_second, max_over_ranks

    # CE_BENCH_REHEARSE=1 (dev only, 1-GPU box): all ranks share cuda:0 and rendezvous over gloo, to exercise the
    # multi-rank control flow where only one GPU exists; its numbers mean nothing
    rehearse = os.environ.get("CE_BENCH_REHEARSE") == "1"
    # CE_BENCH_FORCE_PG=1: take the N > 1 branch at ANY world size -- RCCL bootstrap (init_process_group("nccl")), the
    # device-tensor all_reduce(MAX) of the timing, dist.barrier() and destroy_process_group() then run at world size 1 on a
    # one-GPU box exactly as they will in the driver's 8-GPU job (tests/test_bench_multirank.py; profiles/round3_bench_rccl_1rank.json)
    use_pg = world > 1 or os.environ.get("CE_BENCH_FORCE_PG") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        # an RCCL / rendezvous failure propagates (non-zero exit with the library's own message); nothing here retries
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    hop1, hop2, cfg = S.numpy_hops(case)
    plan = E.make_plan(hop1, hop2, cfg, case["beta"], 1, case["n_prb_grid"], case["n_sym"], dev, wl.get("interp", "linear"))
    rx, pilots = S.torch_inputs(case, n_slots, n_ports, dev, seed=1234 + rank)
    out = E.estimate_with_plan(plan, rx, pilots)                    # allocates the outputs once
    denoiser = None
    if wl.get("denoise"):
        from srsran_ce_pytorch_amd.denoiser import Denoiser, random_weights
        denoiser = Denoiser(random_weights(0), dev)
    torch.cuda.synchronize()

    def step():
        E.estimate_with_plan(plan, rx, pilots, out)
        if denoiser is not None:
            denoiser(out[0])

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    setup_s = time.perf_counter() - t_start      # imports, process group, plan, synthetic inputs resident, first launch (+ the CPU baseline at N = 1)
    elapsed, kernel_ms = time_steps(step, barrier, args.steps, args.warmup,
                                    lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
    elapsed, kernel_ms = max_over_ranks([elapsed, kernel_ms], "cpu" if rehearse else dev)  # measurement only, not data path

    # sanity: the batch really was estimated (finite outputs, CFO in the generated range)
    assert bool(torch.isfinite(out[1]).all()) and bool(torch.isfinite(out[0][-1, -1].real).all())

    bytes_per_slot = n_ports * plan.alg_bytes_per_item + plan.pilot_bytes_per_slot
    bytes_per_launch = n_slots * bytes_per_slot
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    value = aggregate_slots_per_second(n_slots, args.steps, elapsed, world)
    # HBM traffic per launch from the PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE doubled as the
    # MI355X guide prescribes for gfx950); collected once per round and committed under profiles/ together with the git
    # head and a hash of the kernel sources it was taken at.  A profile of OTHER kernel sources than the ones this run
    # was built from says nothing about this run: `traffic` is then null (and `traffic_stale` says why).
    traffic, traffic_src, traffic_head, traffic_stale = None, None, None, None
    for prof in sorted((ROOT / "profiles").glob("round*_summary.json"), reverse=True):   # newest round first
        pj = json.loads(prof.read_text())
        if pj.get("workload", "").startswith(args.workload) and n_slots == wl["slots"] and "hbm_traffic_bytes_per_launch" in pj:
            traffic_src, traffic_head = f"profiles/{prof.name}", pj.get("head")
            if pj.get("kernel_source_sha16") == kernel_source_sha():
                traffic = pj["hbm_traffic_bytes_per_launch"]
            else:
                traffic_stale = "kernel sources changed since the profile was taken"
            break
    line = {
        "metric": "slots/sec (273-PRB PUSCH, 4 Rx)" if n_ports == 4 else f"slots/sec (273-PRB PUSCH, {n_ports} Rx)",
        "value": value, "unit": "slots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "setup_s": setup_s, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearse else ""),
        "config": {"workload": args.workload, "n_prb": 273, "n_sc": plan.n_sc, "n_sym": plan.n_sym, "dmrs_symbols": [2, 11],
                   "layers": 1, "rx_ports": n_ports, "smoothing": wl["smoothing"], "interp": wl.get("interp", "linear"), "slots_per_gpu": n_slots,
                   "global_slots": world * n_slots, "rx_layout": "[slot][port][sym][sc]", "parallelism": f"slot-shard x{world}, no collective",
                   "process_group": ("gloo (rehearsal)" if rehearse else "nccl (RCCL)") if use_pg else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_head": traffic_head,
                     **({"traffic_stale": traffic_stale} if traffic_stale else {}), "lds_bytes_per_workgroup": plan.lds_bytes,
                     "kernel": "ce_estimate_kernel<1,1,2,7,%d>" % (1 if wl["smoothing"] == "filter" else 3 if wl["smoothing"] == "mmse" else 0), "kernel_ms": kernel_ms,
                     "alg_bytes_per_slot": bytes_per_slot, "alg_bytes_per_launch": bytes_per_launch},
    }
    if world == 1 and args.workload == "pusch273_4rx_filter" and not args.no_secondary:
        line["secondary"] = secondary_lines(E, S, plan, case, rx, pilots, out, n_slots, n_ports, dev)
    if denoiser is not None:
        # dominant kernel of this workload: the denoiser; timed on its own (same stream, same resident batch)
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record()
        for _ in range(args.steps):
            denoiser(out[0])
        d1.record()
        torch.cuda.synchronize()
        dn_ms = max_over_ranks([d0.elapsed_time(d1) / args.steps], "cpu" if rehearse else dev)[0]
        flops = n_slots * n_ports * plan.n_sc * plan.n_sym * DENOISE_FLOP_PER_PIXEL
        ach = flops / (dn_ms * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TFLOPS,
                            "traffic": None, "kernel": "ce_denoise_kernel", "kernel_ms": dn_ms, "alg_flop_per_launch": flops,
                            "note": "useful flops of the three 3x3 layers (2->16->16->2); the kernel issues 1.8x that on MFMA (K padding of layer 1, banded layer 3)",
                            "estimation_kernel_ms": kernel_ms - dn_ms}
        line["dtype"] = "f32 estimation + f16 Conv2d (f32 accumulate)"
        line["config"]["extension"] = "Conv2d denoiser, random weights, parity unpinned"
    if cpu is not None:
        line["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
