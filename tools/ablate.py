#!/usr/bin/env python3
"""Dev tool: time kernel variants built with -DCE_ABLATE=<mask> / -DCE_NT_STORE (GPU box only).

Usage: python tools/ablate.py [--slots 8192] [--ports 4] [--smoothing filter] name=flags ...
e.g.   python tools/ablate.py full= nota=-DCE_ABLATE=1 nowr=-DCE_ABLATE=8
Builds /tmp/libce_hip_ablate.so per variant and runs it in a subprocess (fresh process per variant so the
library is reloaded), prints ms per launch and achieved algorithmic GB/s."""
import argparse, json, subprocess, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from srsran_ce_pytorch_amd import estimator as E, synth as S
slots, ports, smoothing, layers = %d, %d, %r, %d
case = S.bench_case(smoothing, layers, seed=1)
h1, h2, cfg = S.numpy_hops(case)
dev = torch.device("cuda:0")
plan = E.make_plan(h1, h2, cfg, case["beta"], layers, 273, 14, dev)
rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
out = E.estimate_with_plan(plan, rx, pil)
torch.cuda.synchronize()
best = min(E.time_with_plan(plan, rx, pil, out, 2, 10) for _ in range(3))
b = slots * (ports * plan.alg_bytes_per_item + plan.pilot_bytes_per_slot)
print(json.dumps(dict(ms=best, gbs=b / best / 1e6, lds=plan.lds_bytes)))
'''

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=8192)
    ap.add_argument("--ports", type=int, default=4)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--smoothing", default="filter")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    from srsran_ce_pytorch_amd import _lib
    for v in a.variants:
        name, _, flags = v.partition("=")
        _lib.build(force=True, extra_flags=flags.split(), out="/tmp/libce_hip_ablate.so")   # all translation units, concurrently
        r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), a.slots, a.ports, a.smoothing, a.layers)], capture_output=True, text=True,
                           env=dict(os.environ, CE_HIP_LIB="/tmp/libce_hip_ablate.so"))   # never touches the shipped library  # stderr (compiler warnings) dropped
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"{name:12s} {flags:28s} {line[-1] if line else r.stderr[-400:]}", flush=True)

if __name__ == "__main__":
    main()
