#!/usr/bin/env python3
"""Dev tool (GPU box): measured deviation of the HIP path from the reference's outputs on every golden fixture."""
import sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from conftest import golden_names, load_fixture, ta_tie_alternatives
from srsran_ce_pytorch_amd import estimator as E
dev = torch.device("cuda:0")
rows = []
for name in golden_names():
    fx = load_fixture(name)
    interp = "cnn" if fx.variant == "C" else "linear"
    g = torch.as_tensor(fx.grids, device=dev)[None]
    out = E.estimate(g, torch.as_tensor(fx.pilots, device=dev), fx.beta, fx.hop1, fx.hop2, fx.config, interp=interp)
    ch = out[0][0].cpu().numpy()
    e_ch = float(np.abs(ch - fx.ref_ch_est).max() / np.abs(fx.ref_ch_est).max())
    sc = np.stack([o[0].cpu().numpy() if o.numel() else np.full(fx.grids.shape[0], np.nan) for o in out[1:]], 1)
    ref = fx.ref_scalars
    rel = lambda i: float(np.nanmax(np.abs(sc[:, i] - ref[:, i]) / np.maximum(np.abs(ref[:, i]), 1e-300)))
    rows.append(dict(fixture=name, variant=fx.variant, ch_est=e_ch, noise=rel(0) if name != "cfg1_25prb_1dmrs_none" else None, rsrp=rel(1), epre=rel(2),
                     ta_equal=bool(np.all(sc[:, 3] == ref[:, 3])), cfo=None if np.isnan(ref[0, 4]) else rel(4)))
    if fx.variant == "N":   # near-tie class (tools/make_ta_neartie.py): equal, or the neighbour bin the reference's own powers allow
        rows[-1]["ta_equal_or_tied_neighbour"] = bool(all(sc[it, 3] == ref[it, 3] or sc[it, 3] in ta_tie_alternatives(fx, it) for it in range(fx.grids.shape[0])))
    print(f"{name:28s} {fx.variant} ch {e_ch:.1e} rsrp {rows[-1]['rsrp']:.1e} epre {rows[-1]['epre']:.1e} ta_equal {rows[-1]['ta_equal']} cfo {rows[-1]['cfo']}")
json.dump(rows, open(ROOT / "gpurun_out" / "parity_report.json", "w"), indent=1)
print("worst ch_est rel-max:", max(r["ch_est"] for r in rows))
tn = [r for r in rows if r["variant"] != "N"]
print(f"TA bit-identical on {sum(r['ta_equal'] for r in tn)} of {len(tn)} reference fixtures (T + C); near-tie class: identical on "
      f"{sum(r['ta_equal'] for r in rows if r['variant'] == 'N')} of {sum(r['variant'] == 'N' for r in rows)}, the rest on the tied neighbour bin: "
      f"{all(r.get('ta_equal_or_tied_neighbour', True) for r in rows)}")
