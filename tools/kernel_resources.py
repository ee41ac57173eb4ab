#!/usr/bin/env python3
"""Dev tool (build container): per-kernel register / spill / occupancy table of the estimation kernels, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py [-DFLAG ...] [unit-substring]
"""
import re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "srsran_ce_pytorch_amd" / "csrc"


def resources(src, flags=()):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}", "-c", str(src),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *flags]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def short(name):
    m = re.search(r"ce_estimate_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
    if m:
        return "<L%s,NH%s,ND%s,KPT%s,F%s>" % m.groups()
    m = re.search(r"ce_narrow_kernelILi(\d)ELi(\d)E", name)
    return "narrow<L%s,NH%s>" % m.groups() if m else name[:40]


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    srcs = [s for s in sorted(CSRC.glob("ce_inst_*.hip")) if not only or any(o in s.name for o in only)]
    sys.path.insert(0, str(ROOT))
    from srsran_ce_pytorch_amd._lib import EXTRA_FLAGS   # each unit with its own flags, as the build applies them
    with ThreadPoolExecutor(8) as pool:
        res = list(pool.map(lambda s: resources(s, list(EXTRA_FLAGS.get(s.name, [])) + flags), srcs))
    print(f"{'kernel':26s} {'VGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>8s} {'occ':>4s}")
    for src, rows in zip(srcs, res):
        print(f"# {src.name}")
        for r in rows:
            print(f"{short(r['name']):26s} {r.get('VGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} "
                  f"{r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
