#!/usr/bin/env python3
"""Program for rocprofv3 to wrap (GPU box): a few launches of the narrow-allocation geometries (tools/perf_cases.py), in a fixed
order, through whatever kernel their plans select (the wave-per-item kernel where it applies).  Prints the order."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
from perf_cases import CASES

WANT = ["L1 2 hops x 1dmrs 12 PRB in 52", "L2 2 hops x 2dmrs 12 PRB in 52", "harness case4: 2 hops x 3 PRB, full-slot hops",
        "harness case8-like: L2 3 PRB 4dmrs", "harness case0: 3 PRB @40, 4dmrs, 52 grid", "L2 25 PRB in 52"]
dev = torch.device("cuda:0")
by_name = {n: (c, i) for n, c, i in CASES}
for name in WANT:
    case, interp = by_name[name]
    h1, h2, cfg = S.numpy_hops(case)
    plan = E.make_plan(h1, h2, cfg, case["beta"], case["n_layers"], case["n_prb_grid"], case["n_sym"], dev, interp)
    rx, pil = S.torch_inputs(case, 8192, 4, dev, 1)
    out = E.estimate_with_plan(plan, rx, pil)
    for _ in range(60):
        E.estimate_with_plan(plan, rx, pil, out)
    torch.cuda.synchronize()
    print(f"{name}: 61 launches, lds={plan.lds_bytes}", flush=True)
    del rx, pil, out
