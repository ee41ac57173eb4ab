#!/usr/bin/env python3
"""Dev tool (GPU box): A/B of whole libraries INSIDE ONE PROCESS, on the same resident buffers.

Launch time is stable to ~0.2 % inside a process and varies by 2-4 % between processes of one lease (where the buffers
land physically), so this resolves differences tools/ab_libs.py (one child process per library) cannot.  Each library is
loaded through its own copy of the host package (ctypes handles and plan caches are per copy).

    python tools/ab_inproc.py [--slots 8192] [--ports 4] [--rounds 5] [--only substring] name=path/to/lib.so ...   ("cur" = the tree's library)
"""
import argparse, importlib, os, shutil, statistics as st, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=8192)
    ap.add_argument("--ports", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="", help="comma-separated substrings of case names")
    ap.add_argument("--check-bits", action="store_true", help="also run every library once into its own outputs and compare them bit for bit with the first library's")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    import torch
    from srsran_ce_pytorch_amd import synth as S, _lib as L0
    from perf_cases import CASES
    tmp = Path(tempfile.mkdtemp(prefix="abpkg_"))
    sys.path.insert(0, str(tmp))
    pk = {}
    for i, spec in enumerate(a.libs):
        name, path = spec.split("=", 1)
        path = str(L0.LIB_PATH if path == "cur" else Path(path).resolve())
        mod = f"abpkg{i}"
        shutil.copytree(ROOT / "srsran_ce_pytorch_amd", tmp / mod, ignore=shutil.ignore_patterns("csrc", "__pycache__"))
        os.environ["CE_HIP_LIB"] = path
        E = importlib.import_module(mod + ".estimator")
        importlib.import_module(mod + "._lib").load()
        pk[name] = E
    dev = torch.device("cuda:0")
    names = list(pk)
    print(f"# {a.slots} slots x {a.ports} ports, one process, {a.rounds} alternating rounds; ms per launch: min (median) per library; last column: "
          f"{names[-1]} vs {names[0]} by medians")
    print(f"{'case':42s}" + "".join(f"{n:>18s}" for n in names))
    for cname, case, interp in CASES:
        if not any(o in cname for o in a.only.split(",")):
            continue
        h1, h2, cfg = S.numpy_hops(case)
        rx, pil = S.torch_inputs(case, a.slots, a.ports, dev, 1)
        plans, outs = {}, {}
        for n, E in pk.items():
            plans[n] = E.make_plan(h1, h2, cfg, case["beta"], case["n_layers"], case["n_prb_grid"], case["n_sym"], dev, interp)
            outs[n] = E.estimate_with_plan(plans[n], rx, pil) if not outs else None
        out = next(o for o in outs.values() if o is not None)
        same = ""
        if a.check_bits:
            ref = [o.clone() for o in pk[names[0]].estimate_with_plan(plans[names[0]], rx, pil)]
            for n in names[1:]:
                got = pk[n].estimate_with_plan(plans[n], rx, pil)
                bad = []
                for nm, g, r in zip(("ch_est", "noise", "rsrp", "epre", "ta", "cfo"), got, ref):
                    gi, ri = (torch.view_as_real(g), torch.view_as_real(r)) if g.is_complex() else (g.view(torch.int64), r.view(torch.int64))
                    if not torch.equal(gi, ri):
                        ne = (gi != ri)
                        d = (g - r).abs().nan_to_num().max().item() / max(r.abs().nan_to_num().max().item(), 1e-30)
                        where = ""
                        if g.is_complex():   # which subcarriers of the first differing item
                            items = ne.flatten(2).any(-1)                       # [B][R]
                            idx = items.nonzero()[0].tolist()
                            scs = ne[idx[0], idx[1]].flatten(1).any(-1).nonzero().flatten()
                            where = f", item {idx}: {scs.numel()} subcarriers, first {scs[:6].tolist()}, last {scs[-3:].tolist()}"
                        bad.append(f"{nm} ({int(ne.sum())} words, rel-max {d:.1e}{where})")
                same += f"  {n}: " + ("bit-identical" if not bad else "DIFFERENT BITS in " + "; ".join(bad))
                del got
            del ref
        t = {n: [] for n in names}
        for r in range(a.rounds):
            for n in (names if r % 2 == 0 else names[::-1]):
                t[n].append(pk[n].time_with_plan(plans[n], rx, pil, out, 1, 5))
        med = {n: st.median(t[n]) for n in names}
        print(f"{cname:42s}" + "".join(f"{min(t[n]):9.3f} ({med[n]:6.3f})" for n in names) + f"   {100 * (med[names[-1]] / med[names[0]] - 1):+.2f} %" + same, flush=True)
        del rx, pil, out
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
