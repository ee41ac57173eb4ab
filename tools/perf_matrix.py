#!/usr/bin/env python3
"""Dev tool (GPU box): kernel time / achieved algorithmic GB/s over the geometries the kernel templates cover."""
import sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
sys.path.insert(0, str(ROOT / "tools"))
from perf_cases import CASES as cases
dev = torch.device("cuda:0")
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ports = int(sys.argv[3]) if len(sys.argv) > 3 else 4
only = sys.argv[2] if len(sys.argv) > 2 else ""   # substring filter on the case name
for name, case, interp in cases:
    if only not in name:
        continue
    h1, h2, cfg = S.numpy_hops(case)
    L = case["n_layers"]
    plan = E.make_plan(h1, h2, cfg, case["beta"], L, case["n_prb_grid"], case["n_sym"], dev, interp)
    rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
    out = E.estimate_with_plan(plan, rx, pil)
    torch.cuda.synchronize()
    cold = min(E.time_with_plan(plan, rx, pil, out, 1, 5) for _ in range(2))
    # short kernels are over in a few ms, before the clocks have settled: keep the GPU busy for ~150 ms, then time 20 launches
    # (bench.py's own warm-up + 20 steps of a 2.7 ms kernel is past that point)
    E.time_with_plan(plan, rx, pil, out, 0, max(5, int(150.0 / cold)))
    ms = min(E.time_with_plan(plan, rx, pil, out, 0, 20) for _ in range(2))
    b = slots * (ports * plan.alg_bytes_per_item + plan.pilot_bytes_per_slot)
    print(f"{name:36s} {ms:8.3f} ms  {b / ms / 1e6:7.0f} GB/s  {slots / ms * 1e3 / 1e6:6.2f} M slots/s  lds={plan.lds_bytes}  (first 10 launches: {cold:.3f} ms)", flush=True)
    del rx, pil, out
