#!/usr/bin/env python3
"""Dev tool (GPU box): does the headline's launch time depend on where the output buffer starts relative to the input?
Times the same plan / inputs with the response written at different byte offsets of one oversized allocation."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
dev = torch.device("cuda:0")
case = S.bench_case("filter", 1)
h1, h2, cfg = S.numpy_hops(case)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, case["n_prb_grid"], 14, dev)
slots, ports = 8192, 4
rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
n = slots * ports * 3276 * 14
big = torch.empty(n + (64 << 20) // 8, dtype=torch.complex64, device=dev)
out0 = E.estimate_with_plan(plan, rx, pil)
print("rx ptr %x  big ptr %x" % (rx.data_ptr(), big.data_ptr()))
for rep in range(2):
    for off_b in [0, 256, 1024, 4096, 16384, 65536, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 17 << 20, 33 << 20]:
        o = big[off_b // 8: off_b // 8 + n].view(slots, ports, 3276, 14, 1)
        out = list(out0)
        out[0] = o
        ms = min(E.time_with_plan(plan, rx, pil, tuple(out), 1, 5) for _ in range(3))
        print(f"offset {off_b:>9d} B: {ms:.3f} ms", flush=True)
