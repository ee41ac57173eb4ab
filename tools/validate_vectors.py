#!/usr/bin/env python3
"""Run every srsRAN `port_channel_estimator` test-vector set of a directory through the HIP estimator (GPU box).

    python tools/validate_vectors.py --dir testvector_outputs [--header .../port_channel_estimator_test_data.h] [--cases 0 4 8]

Same protocol as the reference's `scripts/validation/validate_all.py` (which also keeps working unchanged through
`compat/`): sparse grids from the `<HHff` entry files, pilot axis order searched, comparison at the REs the expected
output lists.  The vectors are not distributed with the reference (git-ignored there) nor with this repo."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from srsran_ce_pytorch_amd import estimator as E, vectors as V


def hip(grid, pilots, beta, hop1, hop2, config):
    out = E.srs_channel_estimator(torch.from_numpy(grid).cuda(), torch.from_numpy(pilots).cuda(), beta, hop1, hop2, config)
    return [o.cpu().numpy() for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--header")
    ap.add_argument("--cases", type=int, nargs="*")
    a = ap.parse_args()
    header = Path(a.header) if a.header else Path(a.dir) / "port_channel_estimator_test_data.h"
    cases = V.parse_test_data_header(header.read_text())
    print(f"{len(cases)} test cases in {header}")
    worst = None
    for c in cases:
        if a.cases and c.idx not in a.cases:
            continue
        try:
            r = V.run_vector_case(c, a.dir, hip)
        except (ValueError, AssertionError, FileNotFoundError) as e:
            print(f"case {c.idx:3d}: skipped ({e})")
            continue
        print(f"case {c.idx:3d}: max {r['max']:.2e}  rms {r['rms']:.2e}  pilots {r['order']}  L={r['layers']}")
        if worst is None or r["max"] > worst["max"]:
            worst = r
    if worst:
        print(f"worst: case {worst['idx']} max {worst['max']:.2e} rms {worst['rms']:.2e}")


if __name__ == "__main__":
    main()
