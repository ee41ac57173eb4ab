#!/usr/bin/env python3
"""Dev tool (GPU box): one batch as K concurrent launches on K streams of ONE device (fork / join with events) against the single
launch -- timed with events on the caller's stream, 20 steps each, alternating."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S

dev = torch.device("cuda:0")
smoothing = sys.argv[1] if len(sys.argv) > 1 else "filter"
n_slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n_ports = int(sys.argv[3]) if len(sys.argv) > 3 else 4
case = S.bench_case(smoothing, 1)
h1, h2, cfg = S.numpy_hops(case)
rx, pil = S.torch_inputs(case, n_slots, n_ports, dev, seed=1)
plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
out = E.estimate_with_plan(plan, rx, pil)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
cur = torch.cuda.current_stream(dev)


def step(k):
    if k == 1:
        E.estimate_with_plan(plan, rx, pil, out)
        return
    ev = torch.cuda.Event()
    ev.record(cur)
    b = [(n_slots * i) // k for i in range(k + 1)]
    for i in range(k):
        st = streams[i]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            E.estimate_with_plan(plan, rx[b[i]:b[i + 1]], pil[b[i]:b[i + 1]], tuple(t[b[i]:b[i + 1]] for t in out))
    for i in range(k):
        cur.wait_stream(streams[i])


def timed(k, n=20):
    for _ in range(3):
        step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(n):
        step(k)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"# {smoothing}, {n_slots} slots x {n_ports} ports; ms per step (events on the caller's stream), three rounds")
for rnd in range(3):
    print("  ".join(f"K={k}: {timed(k):.4f}" for k in (1, 2, 3, 4, 6, 8)), flush=True)
