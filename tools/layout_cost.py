#!/usr/bin/env python3
"""Dev tool (GPU box): the same 273-PRB / 4 Rx batch with the received grid in the reference's dense [sc][sym] order
(symbol fastest) vs the recommended [sym][sc] buffer -- what the input layout costs ("layout-inflated traffic")."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
dev = torch.device("cuda:0")
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
case = S.bench_case("filter", 1, seed=1)
h1, h2, cfg = S.numpy_hops(case)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
rx, pil = S.torch_inputs(case, slots, 4, dev, 1)
out = E.estimate_with_plan(plan, rx, pil)
alg = slots * (4 * plan.alg_bytes_per_item + plan.pilot_bytes_per_slot)
for name, g in (("[sym][sc] buffer (recommended)", rx), ("[sc][sym] dense (reference order)", rx.contiguous())):
    ref = E.estimate_with_plan(plan, g, pil)
    assert torch.equal(ref[0], out[0])                        # the layout never changes a result
    ms = min(E.time_with_plan(plan, g, pil, out, 1, 5) for _ in range(2))
    print(f"{name:36s} strides {tuple(g.stride())}: {ms:.3f} ms / {slots} slots, {alg / ms / 1e6:.0f} GB/s algorithmic, {slots / ms / 1e3:.2f} M slots/s")
