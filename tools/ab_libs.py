#!/usr/bin/env python3
"""Dev tool (GPU box): interleaved A/B of whole libraries over tools/perf_cases.py -- every library is timed in its own
child process (fresh load), the set is repeated ROUNDS times in alternating order, and the per-case MIN and MEDIAN are
reported, so clock / thermal drift of the box does not pass for a difference between the libraries.

    python tools/ab_libs.py [--slots 8192] [--ports 4] [--rounds 3] [--only substring] name=path/to/lib.so ...   ("cur" = the tree's library)
"""
import argparse, json, os, statistics as st, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, json
sys.path[:0] = [%r, %r]
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
from perf_cases import CASES
slots, ports, only = %d, %d, %r
dev = torch.device("cuda:0")
res = {}
for name, case, interp in CASES:
    if only not in name:
        continue
    h1, h2, cfg = S.numpy_hops(case)
    plan = E.make_plan(h1, h2, cfg, case["beta"], case["n_layers"], case["n_prb_grid"], case["n_sym"], dev, interp)
    rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
    out = E.estimate_with_plan(plan, rx, pil)
    torch.cuda.synchronize()
    res[name] = min(E.time_with_plan(plan, rx, pil, out, 1, 5) for _ in range(2))
    del rx, pil, out
print("RESULT " + json.dumps(res))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=8192)
    ap.add_argument("--ports", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    libs = [l.partition("=")[::2] for l in a.libs]
    data = {n: {} for n, _ in libs}
    for r in range(a.rounds):
        for n, path in (libs if r % 2 == 0 else libs[::-1]):
            env = dict(os.environ)
            if path and path != "cur":
                env["CE_HIP_LIB"] = path
            else:
                env.pop("CE_HIP_LIB", None)
            p = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), str(ROOT / "tools"), a.slots, a.ports, a.only)], capture_output=True, text=True, env=env)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(n, "FAILED", p.stderr[-500:])
                continue
            for k, v in json.loads(line[-1][7:]).items():
                data[n].setdefault(k, []).append(v)
    names = [n for n, _ in libs]
    print(f"# {a.slots} slots x {a.ports} ports, {a.rounds} interleaved rounds; ms per launch: min (median) per library")
    print(f"{'case':40s}" + "".join(f"{n:>18s}" for n in names))
    for k in data[names[0]]:
        print(f"{k:40s}" + "".join(f"{min(data[n].get(k, [float('nan')])):9.3f} ({st.median(data[n].get(k, [float('nan')])):6.3f})" for n in names))


if __name__ == "__main__":
    main()
