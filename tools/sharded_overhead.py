#!/usr/bin/env python3
"""Dev tool (GPU box): what `sharding.estimate_sharded` costs next to one `estimate_with_plan` call -- the headline batch whole, and
as 2 / 4 / 8 shards on ONE device (one launch stream each; on a real node the shards sit on different devices)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from srsran_ce_pytorch_amd import estimator as E, sharding as SH, synth as S

dev = torch.device("cuda:0")
case = S.bench_case("filter", 1)
h1, h2, cfg = S.numpy_hops(case)
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rx, pil = S.torch_inputs(case, n_slots, 4, dev, seed=1)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
out = E.estimate_with_plan(plan, rx, pil)
torch.cuda.synchronize()


def timed(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - t0) / n * 1e3


h, w = timed(lambda: E.estimate_with_plan(plan, rx, pil, out))
print(f"one call, {n_slots} slots x 4: host {h:.3f} ms per call, wall {w:.3f} ms per call")
for k in (2, 4, 8):
    rs, ps = SH.split_slots(rx, pil, [0] * k)
    outs = SH.estimate_sharded(rs, ps, case["beta"], h1, h2, cfg, devices=[0] * k)
    torch.cuda.synchronize()
    h, w = timed(lambda: SH.estimate_sharded(rs, ps, case["beta"], h1, h2, cfg, devices=[0] * k, outs=outs))
    print(f"{k} shards on one device ({k} streams): host {h:.3f} ms per call, wall {w:.3f} ms per call")
    del rs, ps, outs
