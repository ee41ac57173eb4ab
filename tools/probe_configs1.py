#!/usr/bin/env python3
"""Dev tool (GPU box): configs[1] (1024 slots x 1 Rx, none) timed on eight fresh allocations of its buffers inside one process."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
dev = torch.device("cuda:0")
case = S.bench_case("none", 1, seed=4321)
h1, h2, cfg = S.numpy_hops(case)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
keep = []
for trial in range(8):
    if trial == 4:
        keep.append(torch.empty(6 << 30, dtype=torch.uint8, device=dev))   # shift where later allocations land
    rx, pil = S.torch_inputs(case, 1024, 1, dev, seed=trial)
    out = E.estimate_with_plan(plan, rx, pil)
    E.time_with_plan(plan, rx, pil, out, 0, 300)
    ms = [E.time_with_plan(plan, rx, pil, out, 0, 20) for _ in range(5)]
    print(f"trial {trial}: rx {rx.data_ptr():x} out {out[0].data_ptr():x}  {min(ms):.4f} .. {max(ms):.4f} ms", flush=True)
    keep.append((rx, pil, out))
