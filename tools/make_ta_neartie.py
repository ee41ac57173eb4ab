#!/usr/bin/env python3
"""Pin the time-alignment arg-max on near-ties against the REAL reference (build container only).

On narrow bands the 4096-point IFFT of the reference (T:670-698) has a main lobe tens of bins wide; when the delay
falls midway between two bins, the two largest powers differ by float32 rounding only and the arg-max is decided by
the transform's rounding noise (torch.fft.ifft in complex64 for the reference, numpy for the oracle, a pruned radix-16
transform for the HIP kernel).  This tool

  1. replays the TA disagreements the differential fuzzer logged on the GPU box (gpurun_out/fuzz_*.log) and searches
     seeded random narrow-band slots for more near-ties,
  2. runs the real ce_rule_tensorized on each, capturing the IFFT it computes (a recording wrapper around
     torch.fft.ifft, nothing of the reference is changed) to store, per hop, the arg-max bin the reference chose and
     the power around it,
  3. writes tests/golden/ta_neartie_*.npz (variant "N": inputs, the six reference outputs, ta_bin[item][hop],
     ta_pw[item][hop][3] = reference power at bin-1, bin, bin+1 on the side it chose) and prints how often the numpy
     oracle disagrees with the reference on the set.

The test protocol for variant "N" (tests/conftest.py::check_ta_neartie): the TA must equal the reference's, or differ
by ONE bin in ONE hop towards a neighbour whose reference power is within TA_TIE_RATIO of the winner's.
"""
from __future__ import annotations

import json
import re
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), "/root/reference/src"]

import ce_rule_tensorized as REF_T   # noqa: E402
import ce_oracle as O                # noqa: E402
from srsran_ce_pytorch_amd import synth as S   # noqa: E402

HALF = 144


class IfftTap:
    """Records the output of every torch.fft.ifft call made while active (the reference calls it once per hop, T:679)."""

    def __enter__(self):
        self.calls = []
        self._orig = torch.fft.ifft

        def rec(*a, **k):
            out = self._orig(*a, **k)
            self.calls.append(out.detach().clone())
            return out

        torch.fft.ifft = rec
        return self

    def __exit__(self, *exc):
        torch.fft.ifft = self._orig


def _ref_hop(ha):
    return REF_T.HopConfig(torch.as_tensor(ha.DMRSsymbols), torch.as_tensor(ha.DMRSREmask), ha.PRBstart, ha.nPRBs,
                           torch.as_tensor(ha.maskPRBs), ha.startSymbol, ha.nAllocatedSymbols)


def run_reference(b, grid):
    cfg = REF_T.EstimatorConfig(b.config.scs, torch.as_tensor(b.config.CyclicPrefixDurations), b.config.Smoothing, b.config.CFOCompensate)
    with torch.no_grad(), IfftTap() as tap:
        out = REF_T.srs_channel_estimator(torch.as_tensor(grid), torch.as_tensor(b.pilots), b.beta, _ref_hop(b.hop1), _ref_hop(b.hop2), cfg)
    ch = out[0].numpy()
    scal = np.array([float(x) for x in out[1:5]] + [float(out[5]) if out[5].numel() else float("nan")], np.float64)
    bins, pws, margins = [], [], []
    for ir in tap.calls:                                   # the reference's own arithmetic on its own IFFT (T:680-696)
        lp = torch.sum(torch.abs(ir) ** 2, dim=1)
        head, tail = lp[:HALF], lp[-HALF:]
        md, idl = torch.max(head, dim=0)
        ma, ia = torch.max(tail, dim=0)
        if float(md) >= float(ma):
            side, i = head, int(idl)
            bins.append(i)
        else:
            side, i = tail, int(ia)
            bins.append(-(HALF - i))
        s = side.numpy().astype(np.float64)
        # neighbours among the examined bins; bins 4095 (advance side's last) and 0 (delay side's first) are neighbours too
        below = s[i - 1] if i > 0 else (float(tail[HALF - 1]) if side is head else -1.0)
        above = s[i + 1] if i + 1 < HALF else (float(head[0]) if side is tail else -1.0)
        pws.append([below, s[i], above])
        rest = np.delete(s, i)
        margins.append(float((s[i] - rest.max()) / s[i]))
    return ch, scal, bins, pws, margins


def logged_cases():
    out = []
    for log in sorted((ROOT / "gpurun_out").glob("fuzz_*.log")):
        for line in log.read_text().splitlines():
            m = re.match(r"\[(\d+)\] MISMATCH .*time alignment .* :: (\{.*\})$", line)
            if m:
                case = json.loads(m.group(2))
                if case.pop("scattered", False) or case.pop("interp", "linear") != "linear" or case["smoothing"] == "mmse":
                    continue
                case.pop("layout_ref", None)
                case["name"] = f"ta_neartie_log{m.group(1)}"
                out.append(case)
    return out


def random_cases(n_want, seed=2024, margin=2e-6):
    """Seeded narrow-band slots whose reference arg-max wins by less than `margin` (relative power) on some item/hop."""
    rng = np.random.default_rng(seed)
    masks = [S.TYPE1_CDM0, S.TYPE1_CDM1, S.TYPE2_CDM0, [1, 0, 0, 0] * 3, [1, 0, 0, 0, 0, 0] * 2, [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0]]
    found, tried, n_wide = [], 0, 0
    while len(found) < n_want and tried < 20000:
        tried += 1
        want_wide = n_wide < n_want // 2                   # half the set with >= 12 pilots (rarer: the lobe is narrower)
        grid = int(rng.choice([6, 25, 52]))
        n_prbs = int(rng.integers(2, 7)) if want_wide else int(rng.integers(1, 3))
        n_prbs = min(n_prbs, grid)
        n_hops = int(rng.choice([1, 1, 2]))
        mask = masks[rng.integers(len(masks))]
        hops = []
        for h in range(n_hops):
            lo, hi = (0, 14) if n_hops == 1 else ((0, 7) if h == 0 else (7, 14))
            dm = sorted(rng.choice(np.arange(lo, hi), size=int(rng.integers(1, 3)), replace=False).tolist())
            hops.append(S.hop_spec(dm, int(rng.integers(0, grid - n_prbs + 1)), n_prbs, lo, hi - lo, [mask]))
        case = S.case_spec(f"ta_neartie_rand{len(found):02d}", grid, hops, n_layers=int(rng.choice([1, 1, 2])),
                           smoothing=str(rng.choice(["none", "filter", "mean"])), scs=float(rng.choice([15e3, 30e3])),
                           seed=int(rng.integers(1 << 30)), delay_ns=float(rng.uniform(0, 500)), cfo_hz=float(rng.uniform(50, 300)))
        try:
            b = S.build_case(case, 2)
            n_pil = b.pilots.shape[0]
            if n_pil < 3 or (want_wide and n_pil < 12) or (not want_wide and n_pil >= 12):
                continue                                   # 1-2 pilots: |IFFT| is flat / periodic, the arg-max is arbitrary by construction
            worst = min(min(run_reference(b, b.grids[it])[4]) for it in range(2))
        except Exception:
            continue
        if worst < margin:
            found.append(case)
            n_wide += int(n_pil >= 12)
    print(f"random search: {len(found)} near-tie cases in {tried} draws (margin < {margin:g})")
    return found


def main():
    out_dir = ROOT / "tests" / "golden"
    manifest_p = out_dir / "MANIFEST.json"
    manifest = json.loads(manifest_p.read_text())
    cases = logged_cases() + random_cases(16)
    n_items_total = n_oracle_diff = 0
    for f in out_dir.glob("ta_neartie_*.npz"):
        f.unlink()
    for case in cases:
        b = S.build_case(case, 2)
        if b.pilots.shape[0] < 3:
            print(f"{case['name']}: {b.pilots.shape[0]} pilot(s) -- arg-max arbitrary by construction, skipped")
            continue
        cols = sorted({s for h in case["hops"] for s in h["dmrs_symbols"]})
        grids = np.zeros_like(b.grids)
        grids[:, :, cols] = b.grids[:, :, cols]
        chs, scs, bins, pws = [], [], [], []
        worst = 1.0
        for it in range(2):
            ch, sc, bn, pw, mg = run_reference(b, grids[it])
            chs.append(ch); scs.append(sc); bins.append(bn); pws.append(pw)
            worst = min(worst, min(mg))
            ora = O.srs_channel_estimator(grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config)
            n_items_total += 1
            n_oracle_diff += int(float(ora[4]) != sc[3])
        np.savez_compressed(out_dir / f"{case['name']}.npz", case_json=np.array(json.dumps(case)), variant=np.array("N"),
                            pilots=b.pilots, grid_cols=grids[:, :, cols], cols=np.array(cols, np.int64),
                            ref_ch_est=np.stack(chs), ref_scalars=np.stack(scs),
                            ta_bin=np.array(bins, np.int64), ta_pw=np.array(pws, np.float64))
        manifest[case["name"]] = dict(variant="N", n_items=2, smallest_relative_margin_of_the_reference_argmax=worst,
                                      scalars=["noise", "rsrp", "epre", "time_alignment", "cfo_hz(nan=not estimated)"])
        print(f"{case['name']:26s} pilots={b.pilots.shape[0]:3d} hops={len(case['hops'])} ref bins={bins} smallest margin={worst:.2e}")
    manifest_p.write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    print(f"oracle (numpy IFFT) vs reference (torch.fft.ifft complex64): TA differs on {n_oracle_diff} of {n_items_total} items")


if __name__ == "__main__":
    main()
