"""Static checks on the gfx950 code of every estimation-kernel instantiation (cross-compiled here, no GPU needed):

* no flat_load / flat_store: every LDS and global access must carry its address space.  (A lambda or helper the
  compiler leaves out of line receives its LDS pointers as flat addresses; an index that is merely out of range then
  becomes a memory-aperture fault instead of a harmless LDS read -- this happened once.)
* the register-path kernels of the parity-pinned feature sets (one layer, none / mean / RC filter) keep everything in
  registers, except the narrowest shapes built for one more workgroup per CU by choice (ce_min_waves: measured trade)."""
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "srsran_ce_pytorch_amd" / "csrc"
NARROW = re.compile(r"^(_ZN12_GLOBAL__N_116ce_narrow_kernelILi(\d)ELi(\d)E\w+):\s*;.*?\n(.*?)s_endpgm", re.S | re.M)
KERNEL = re.compile(r"^(_ZN12_GLOBAL__N_118ce_estimate_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E\w+):\s*;.*?\n(.*?)s_endpgm", re.S | re.M)


def _asm(src: Path) -> str:
    from srsran_ce_pytorch_amd import _lib
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}",
           *_lib.EXTRA_FLAGS.get(src.name, []), "-S", "--cuda-device-only", str(src), "-o", "-"]   # the unit's own flags, as the build applies them
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def test_no_flat_addressing_and_no_spills_on_the_pinned_register_path():
    srcs = sorted(CSRC.glob("ce_inst_*.hip"))
    assert len(srcs) == 8
    with ThreadPoolExecutor(len(srcs)) as pool:
        texts = list(pool.map(_asm, srcs))
    seen, seen_narrow = 0, []
    for src, text in zip(srcs, texts):
        for m in KERNEL.finditer(text):
            L, NH, ND, KPT, FEAT = (int(x) for x in m.group(2, 3, 4, 5, 6))
            body = m.group(7)
            seen += 1
            assert not re.search(r"\bflat_(load|store)", body), f"{src.name} <{L},{NH},{ND},{KPT},{FEAT}> uses flat addressing"
            # built for one more workgroup per CU than their registers allow without a 1-3 register spill (ce_min_waves: measured)
            by_choice = ND * KPT <= 2
            if ND > 0 and FEAT in (0, 1) and not by_choice:
                assert not re.search(r"\bscratch_(load|store)", body), f"{src.name} <{L},{NH},{ND},{KPT},{FEAT}> spills to scratch"
        for m in NARROW.finditer(text):   # the wave-per-item kernels (ce_narrow_kernel.h): LDS / global address spaces only, too
            seen_narrow.append((int(m.group(2)), int(m.group(3))))
            assert not re.search(r"\bflat_(load|store)", m.group(4)), f"{src.name} narrow <{m.group(2)},{m.group(3)}> uses flat addressing"
            assert m.group(4).count("s_barrier") == 1, "the wave-per-item kernel has exactly one workgroup barrier (dead waves leave after it)"
    assert seen == 87, seen
    assert sorted(seen_narrow) == [(l, h) for l in range(1, 5) for h in (1, 2)], seen_narrow
