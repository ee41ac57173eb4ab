"""The drop-in boundary used the way a C/C++ host would: tests/abi_consumer.cpp -- no Python, no torch, buffers from hipMalloc, the null stream --
is built with hipcc against include/ce_hip.h + libce_hip.so, fed fixtures of the real reference through a file, and must give (a) the reference's
outputs within the usual tolerance and (b) the bits the Python host glue gives for the same input."""
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import check_outputs, load_fixture
from srsran_ce_pytorch_amd import _lib, estimator as E

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
SMOOTH = {"none": 0, "mean": 1, "filter": 2}


@pytest.fixture(scope="module")
def consumer(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path_factory.mktemp("abi") / "abi_consumer"
    csrc = _lib.LIB_PATH.parent
    p = subprocess.run([hipcc, "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "abi_consumer.cpp"), f"-L{csrc}", "-lce_hip", f"-Wl,-rpath,{csrc}", "-o", str(exe)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def _write_case(path, fx, sym_major, interp):
    case, n_items = fx.case, fx.grids.shape[0]
    hops = [fx.hop1] + ([fx.hop2] if len(case["hops"]) > 1 else [])
    n_re, n_dm, L = fx.pilots.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<12i", case["n_prb_grid"], case["n_sym"], L, len(hops), SMOOTH[case["smoothing"]], int(case["cfo_compensate"]), interp,
                            1, n_items, n_re, n_dm, int(sym_major)))
        cp = np.zeros(14)
        cp[:len(fx.config.CyclicPrefixDurations)] = np.asarray(fx.config.CyclicPrefixDurations, np.float64)[:14]
        f.write(struct.pack("<17d", float(case["scs"]), float(fx.beta), float(case.get("cnn_alpha", 0.0)), *cp))
        for h in hops:
            dm = np.zeros(14, np.uint8)
            dm[:len(h.DMRSsymbols)] = np.asarray(h.DMRSsymbols, np.uint8)
            f.write(dm.tobytes())
            m = np.asarray(h.DMRSREmask, bool).reshape(12, -1)
            cols = [int(sum(1 << r for r in range(12) if m[r, c])) if c < m.shape[1] else 0 for c in range(2)]
            f.write(struct.pack("<2H4i", cols[0], cols[1], int(h.PRBstart), int(h.nPRBs), int(h.startSymbol), int(h.nAllocatedSymbols)))
            f.write(np.asarray(h.maskPRBs, np.uint8).tobytes())
        g = fx.grids.astype(np.complex64)                                     # [items = ports][sc][sym]
        f.write((g.transpose(0, 2, 1) if sym_major else g).copy().tobytes())
        f.write(fx.pilots.astype(np.complex64).tobytes())


@pytest.mark.parametrize("name,interp,sym_major", [("pusch273_filter_L1", 0, True), ("case4like_fullslot_hops", 0, False), ("layers4_6prb", 0, True),
                                                   ("dmrs5_2hop_layers4_20prb", 0, False), ("sym13_25prb_nocfo", 0, True), ("cnn_type2_3prb", 1, False)])
def test_a_host_without_python_gets_the_same_bits(consumer, tmp_path, name, interp, sym_major):
    fx = load_fixture(name)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    _write_case(src, fx, sym_major, interp)
    p = subprocess.run([str(consumer), str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    raw = dst.read_bytes()
    n_items, n_sc, n_sym, L = fx.grids.shape[0], fx.grids.shape[1], fx.grids.shape[2], fx.pilots.shape[2]
    cfo_estimated = struct.unpack_from("<i", raw, 0)[0]
    ch = np.frombuffer(raw, np.complex64, n_items * n_sc * n_sym * L, 4).reshape(n_items, n_sc, n_sym, L)
    sc = np.frombuffer(raw, np.float64, 5 * n_items, 4 + ch.nbytes).reshape(5, n_items)
    # (a) the reference's outputs
    for it in range(n_items):
        got = [sc[0][it], sc[1][it], sc[2][it], sc[3][it], sc[4][it] if cfo_estimated else np.nan]
        check_outputs(ch[it], got, fx.ref_ch_est[it], fx.ref_scalars[it], 2e-5, 2e-5, f"{name}[{it}] through the C ABI")
    # (b) the Python host glue's bits for the same input and layout
    dev = torch.device("cuda:0")
    g = torch.as_tensor(fx.grids, device=dev)[None]
    if sym_major:
        g = g.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    out = E.estimate(g, torch.as_tensor(fx.pilots, device=dev), fx.beta, fx.hop1, fx.hop2, fx.config, interp="cnn" if interp else "linear")
    torch.cuda.synchronize()
    assert np.array_equal(out[0][0].cpu().numpy().view(np.float32), ch.view(np.float32))
    for j in range(4):
        assert np.array_equal(out[1 + j][0].cpu().numpy().view(np.int64), sc[j].view(np.int64))
    if cfo_estimated:
        assert np.array_equal(out[5][0].cpu().numpy().view(np.int64), sc[4].view(np.int64))
