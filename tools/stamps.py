#!/usr/bin/env python3
"""Dev tool (GPU box): build with -DCE_STAMPS and print where a workgroup's time goes.
Stamps are wall_clock64() (100 MHz) of thread 0 at stage boundaries; shares, not run times."""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
csrc = ROOT / "srsran_ce_pytorch_amd" / "csrc"
flags = sys.argv[1:] 
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT/'include'}", f"-I{csrc}", "-DCE_STAMPS=1",
                "-o", "/tmp/libce_hip_stamps.so", str(csrc / "ce_api.hip"), str(csrc / "ce_denoise.hip"), str(csrc / "ce_kernels.hip")] + flags, check=True)
import os
os.environ["CE_HIP_LIB"] = "/tmp/libce_hip_stamps.so"      # never overwrite the shipped library with a diagnostic build
import torch
from srsran_ce_pytorch_amd import estimator as E, synth as S, _lib
lib = _lib.load()
lib.ce_debug_set_stamps.argtypes = [C.c_void_p]
names = ["init", "load", "cfo", "rot", "ls", "smooth", "resid", "ta", "epilog", "Hbuild", "write"]
dev = torch.device("cuda:0")
for slots, ports in [(64, 4), (8192, 4)]:
    case = S.bench_case("filter", 1, seed=1)
    h1, h2, cfg = S.numpy_hops(case)
    plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
    rx, pil = S.torch_inputs(case, slots, ports, dev, 1)
    n = slots * ports
    st = torch.zeros((n, 16), dtype=torch.int64, device=dev)
    lib.ce_debug_set_stamps(st.data_ptr())
    out = E.estimate_with_plan(plan, rx, pil)
    torch.cuda.synchronize()
    E.estimate_with_plan(plan, rx, pil, out)
    torch.cuda.synchronize()
    t = st.cpu().numpy().astype(np.float64) * 0.01   # us
    d = np.diff(t[:, :11], axis=1)
    print(f"--- {n} items: median us per stage (total {np.median(t[:,10]-t[:,0]):.1f} us; kernel span {(t[:,10].max()-t[:,0].min()):.0f} us)")
    print("  ".join(f"{nm}={np.median(d[:, i]):.2f}" for i, nm in enumerate(names[1:])))
    if t[:, 11].max() > 0:
        print(f"  smoothing detail: conv-phase1 done (wave 0) +{np.median(t[:,11]-t[:,4]):.2f}  virtual pilots done (wave 3) +{np.median(t[:,12]-t[:,4]):.2f}  after barrier +{np.median(t[:,13]-t[:,4]):.2f}  stage end +{np.median(t[:,5]-t[:,4]):.2f}")
