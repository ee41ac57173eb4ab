#!/usr/bin/env python3
"""Dev tool (GPU box): build with -DCE_STAMPS and print where a workgroup's time goes.

    python tools/stamps.py [case-substring ...] [-DFLAG ...]      (cases: tools/perf_matrix.py's table)

Stamps are wall_clock64() (100 MHz) of thread 0 at stage boundaries; shares, not run times."""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
sizes = [(64, 4), (8192, 4)]
args = []
for a in sys.argv[1:]:
    if a.startswith("--sizes="):                      # e.g. --sizes=1024x1,8192x4
        sizes = [tuple(int(v) for v in t.split("x")) for t in a.split("=", 1)[1].split(",")]
    else:
        args.append(a)
flags = [a for a in args if a.startswith("-")]
only = [a for a in args if not a.startswith("-")] or ["L1 2dmrs filter"]
from srsran_ce_pytorch_amd import _lib as _L
_L.build(force=True, extra_flags=["-DCE_STAMPS=1"] + flags, out="/tmp/libce_hip_stamps.so")
os.environ["CE_HIP_LIB"] = "/tmp/libce_hip_stamps.so"      # never overwrite the shipped library with a diagnostic build
import importlib
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S, _lib
importlib.reload(_lib)
lib = _lib.load()
lib.ce_debug_set_stamps.argtypes = [C.c_void_p]
from perf_cases import CASES
names = ["load", "cfo", "rot", "ls", "smooth", "resid", "-", "epilog", "Hbuild", "write", "ta"]
dev = torch.device("cuda:0")
for name, case, interp in CASES:
    if not any(o in name for o in only):
        continue
    for slots, ports in sizes:
        h1, h2, cfg = S.numpy_hops(case)
        plan = E.make_plan(h1, h2, cfg, case["beta"], case["n_layers"], case["n_prb_grid"], case["n_sym"], dev, interp)
        rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
        n = slots * ports
        st = torch.zeros((n, 16), dtype=torch.int64, device=dev)
        lib.ce_debug_set_stamps(st.data_ptr())
        out = E.estimate_with_plan(plan, rx, pil)
        torch.cuda.synchronize()
        E.estimate_with_plan(plan, rx, pil, out)
        torch.cuda.synchronize()
        t = st.cpu().numpy().astype(np.float64) * 0.01   # us
        ta0, ta1, ta2 = t[:, 9].copy(), t[:, 14].copy(), t[:, 15].copy()   # TA start (last hop), after the two radix-16 passes, after the bin sums
        t[:, 9] = t[:, 8]
        d = np.diff(t[:, :12], axis=1)
        print(f"--- {name}: {n} items: median us per stage (item total {np.median(t[:,11]-t[:,0]):.1f} us, to end of writer {np.median(t[:,10]-t[:,0]):.1f}; kernel span {(t[:,11].max()-t[:,0].min()):.0f} us)")
        print("  ".join(f"{nm}={np.median(d[:, i]):.2f}" for i, nm in enumerate(names)), flush=True)
        ta_end = np.where(t[:, 11] > ta2, np.where(t[:, 6] > ta2, np.minimum(t[:, 6], t[:, 11]), t[:, 11]), t[:, 6])
        print(f"  time alignment (last hop): radix-16 passes {np.median(ta1 - ta0):.2f} us, bin sums {np.median(ta2 - ta1):.2f} us, arg-max + seconds {np.median(ta_end - ta2):.2f} us")
        if "-DCE_STAMP_STARTUP" in flags:   # slots 14 / 15: after the plan fields + pilot-load issue, after the plan / twiddle copies
            e = t[:, 13]
            print(f"  start-up: workgroup entry after kernel start: median {np.median(e - e.min()):.2f} us, p95 {np.percentile(e - e.min(), 95):.2f}, max {(e - e.min()).max():.2f}; "
                  f"entry -> plan fields in SGPRs + pilot loads issued {np.median(ta1 - e):.2f} (p95 {np.percentile(ta1 - e, 95):.2f}); "
                  f"-> plan / twiddle copies in LDS {np.median(ta2 - ta1):.2f} (p95 {np.percentile(ta2 - ta1, 95):.2f}); -> first stage stamp {np.median(t[:, 0] - ta2):.2f}")
        print(f"  kernel entry -> first stage stamp (arguments, plan / twiddle / table copies to LDS, pilot loads issued): median {np.median(t[:, 0] - t[:, 13]):.2f} us, p95 {np.percentile(t[:, 0] - t[:, 13], 95):.2f} us")
        # residency: which CU each workgroup ran on (HW_ID bits 8-15: CU / SH / SE, XCC_ID), how many were resident
        # on a CU on average, and how long a CU waited between one workgroup's last stamp and the next one's first
        hw = st.cpu().numpy()[:, 12]
        cu = ((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)
        span = t[:, 11].max() - t[:, 0].min()
        res, gaps = [], []
        for c in np.unique(cu):
            m = cu == c
            s0, e0 = np.sort(t[m, 13]), np.sort(t[m, 11])
            res.append((t[m, 11] - t[m, 13]).sum() / span)
            k = int(round(res[-1] + 0.5)) or 1
            if len(s0) > 2 * k:
                # with k slots busy, the i-th start follows the (i-k)-th end
                gaps.extend((s0[k:] - e0[:-k]).tolist())
        print(f"  {len(np.unique(cu))} CUs seen; resident workgroups per CU: mean {np.mean(res):.2f} (min {np.min(res):.2f}, max {np.max(res):.2f}); "
              f"end-of-TA -> next start on the same CU: median {np.median(gaps) if gaps else float('nan'):.2f} us; "
              f"mean item {np.mean(t[:,11]-t[:,0]):.1f} us, p95 {np.percentile(t[:,11]-t[:,0], 95):.1f} us", flush=True)
        del rx, pil, out
