#!/usr/bin/env python3
"""Dev tool (GPU box): per-call latency of small batches (launch-bound regime): direct ctypes call vs HIP-graph replay."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
dev = torch.device("cuda:0")
case = S.bench_case("filter", 1, seed=3)
h1, h2, cfg = S.numpy_hops(case)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
# host side of one call, nothing waited for: plan-cache lookup alone, the ctypes launch alone, the whole estimate()
rx, pil = S.torch_inputs(case, 1, 4, dev, seed=1)
out = E.estimate_with_plan(plan, rx, pil)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
t1 = time.perf_counter()
for _ in range(n):
    E.estimate_with_plan(plan, rx, pil, out)
t2 = time.perf_counter()
torch.cuda.synchronize()
t3 = time.perf_counter()
for _ in range(n):
    E.estimate(rx, pil, case["beta"], h1, h2, cfg, out=out)
t4 = time.perf_counter()
torch.cuda.synchronize()
print(f"host us per call (1 slot x 4 ports, async): plan-cache hit {(t1 - t0) / n * 1e6:.1f}, estimate_with_plan {(t2 - t1) / n * 1e6:.1f}, "
      f"estimate() {(t4 - t3) / n * 1e6:.1f}")
for slots, ports in ((1, 1), (1, 4), (4, 4), (16, 4), (64, 4)):
    rx, pil = S.torch_inputs(case, slots, ports, dev, seed=1)
    out = E.estimate_with_plan(plan, rx, pil)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        E.estimate_with_plan(plan, rx, pil, out)
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / n * 1e6
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        E.estimate_with_plan(plan, rx, pil, out)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        E.estimate_with_plan(plan, rx, pil, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n * 1e6
    one = time.perf_counter(); g.replay(); torch.cuda.synchronize(); rt = (time.perf_counter() - one) * 1e6
    print(f"{slots:3d} slots x {ports} ports: back-to-back {direct:7.1f} us/call direct, {graph:7.1f} us/call graph replay; one replay + sync {rt:7.1f} us")
