"""The kernel tiers against each other: the same slots through the plan's default kernel and through the alternative code
paths its tuning knobs select (environment variables read when a plan is created -- by the DIAGNOSTIC build of the library
only, csrc/libce_hip_knobs.so = -DCE_TUNING_KNOBS, loaded through CE_HIP_LIB; the shipped libce_hip.so, which the default
run uses, never reads the environment: tests/test_host_and_abi.py) -- the re-read path
instead of the register path, the widest register tier instead of the band's own, the full first TA pass instead of the
collapsed one, DM-RS symbols re-read instead of parked in the LDS, one TA transform at a time.  Knobs that only move data
must give bit-identical results; knobs that change a summation order must agree to rounding.  Each knob runs in a fresh
child process (plans are cached per process)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from srsran_ce_pytorch_amd import _lib, synth as S

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

CASES = [
    S.case_spec("tier_2hop_200prb", 273, [S.hop_spec([1, 5], 0, 200, 0, 7), S.hop_spec([8, 12], 73, 200, 7, 7)], seed=901),
    S.case_spec("tier_2hop_3dmrs_150prb", 273, [S.hop_spec([0, 3, 6], 10, 150, 0, 7), S.hop_spec([7, 10, 13], 100, 150, 7, 7)], seed=902),
    S.case_spec("tier_2hop_3dmrs_200prb", 273, [S.hop_spec([0, 3, 6], 0, 200, 0, 7), S.hop_spec([7, 10, 13], 73, 200, 7, 7)], seed=907),
    S.case_spec("tier_25prb", 52, [S.hop_spec([2, 11], 10, 25)], seed=903),
    S.case_spec("tier_2hop_12prb", 52, [S.hop_spec([2], 3, 12, 0, 7), S.hop_spec([9], 30, 12, 7, 7)], seed=904),
    S.case_spec("tier_L2_40prb", 106, [S.hop_spec([2, 11], 20, 40)], n_layers=2, seed=905),
    S.case_spec("tier_2hop_L2_12prb", 52, [S.hop_spec([1, 5], 3, 12, 0, 7), S.hop_spec([8, 12], 30, 12, 7, 7)], n_layers=2, seed=906),
    # 4 (and 3) layers x 2 wide hops: no LDS for a second set of TA residue blocks, the last hop's second set lies over the first hop's P (plan: ta_over_p; CE_TA_LP1 = one layer at a time)
    S.case_spec("tier_2hop_L4_136prb", 273, [S.hop_spec([1, 5], 0, 136, 0, 7, [S.TYPE1_CDM0, S.TYPE1_CDM1]), S.hop_spec([8, 12], 137, 136, 7, 7, [S.TYPE1_CDM0, S.TYPE1_CDM1])], n_layers=4, seed=908),
    S.case_spec("tier_2hop_L3_120prb", 273, [S.hop_spec([0, 3, 6], 20, 120, 0, 7, [S.TYPE1_CDM0, S.TYPE1_CDM1]), S.hop_spec([7, 10, 13], 150, 120, 7, 7, [S.TYPE1_CDM0, S.TYPE1_CDM1])], n_layers=3, seed=909),
]

CHILD = r'''
import json, sys
sys.path[:0] = [%r, %r]
import numpy as np, torch
from srsran_ce_pytorch_amd import estimator as E, synth as S
cases = json.loads(sys.argv[1])
out = {}
for c in cases:
    b = S.build_case(c, 2)
    g = torch.as_tensor(b.grids, device="cuda:0")[None]
    r = E.estimate(g, torch.as_tensor(b.pilots, device="cuda:0"), b.beta, b.hop1, b.hop2, b.config)
    torch.cuda.synchronize()
    out[c["name"] + "/ch"] = r[0][0].cpu().numpy()
    for k, t in zip(("noise", "rsrp", "epre", "ta", "cfo"), r[1:]):
        out[c["name"] + "/" + k] = t[0].cpu().numpy() if t.numel() else np.zeros(0)
np.savez(sys.argv[2], **out)
''' % (str(ROOT), str(ROOT / "tests"))


def _run(tmp_path, knob, also=()):
    env = dict(os.environ)
    env.pop("CE_HIP_LIB", None)
    if knob:
        if not _lib.KNOBS_LIB_PATH.exists():
            _lib.build(force=True)
        for k in (knob,) + tuple(also):
            env[k] = "1"
        env["CE_HIP_LIB"] = str(_lib.KNOBS_LIB_PATH)
    dst = tmp_path / f"tier_{knob or 'default'}{'_'.join(also)}.npz"
    p = subprocess.run([sys.executable, "-c", CHILD, json.dumps(CASES), str(dst)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return np.load(dst)


@pytest.fixture(scope="module")
def shipped_results(tmp_path_factory):
    """The shipped library's own choice of kernel for every case (narrow allocations: the wave-per-item kernel)."""
    return _run(tmp_path_factory.mktemp("tiers"), None)


@pytest.fixture(scope="module")
def default_results(tmp_path_factory):
    """The workgroup-per-item kernels' default path for every case (CE_NO_NARROW): what their knobs are compared with."""
    return _run(tmp_path_factory.mktemp("tiers_wg"), "CE_NO_NARROW")


def test_wave_per_item_kernel_agrees_with_the_workgroup_per_item_kernels(shipped_results, default_results):
    """ce_narrow_kernel.h against ce_estimate_kernel.h on the narrow cases: same arithmetic per stage, sums in another
    order (wave sums / the TA's cross-row reduce-scatter) -- agreement to rounding, time alignment identical."""
    from srsran_ce_pytorch_amd import estimator as E
    narrow = 0
    for c in CASES:
        n = c["name"]
        h1, h2, cfg = S.numpy_hops(c)
        narrow += E.derive_host(h1, h2, cfg, c["beta"], c["n_layers"], c["n_prb_grid"], c["n_sym"]).narrow   # the shipped plan's choice
        ch0, ch1 = default_results[n + "/ch"], shipped_results[n + "/ch"]
        assert np.abs(ch0 - ch1).max() <= 2e-6 * np.abs(ch0).max(), n
        assert np.array_equal(default_results[n + "/ta"], shipped_results[n + "/ta"]), f"{n} time alignment"
        for k in ("noise", "rsrp", "epre", "cfo"):
            a, b = default_results[n + "/" + k], shipped_results[n + "/" + k]
            assert np.allclose(a, b, rtol=2e-6, atol=1e-9 if k != "cfo" else 1e-3), f"{n} {k}: {a} vs {b}"
    assert narrow >= 2, "the narrow two-hop / multi-layer cases of the set should run on the wave-per-item kernel by default"


# knob -> results must be bit-identical to the default path's (the knob only changes where data waits or how a transform is pruned?)
@pytest.mark.parametrize("knob,bitwise", [("CE_NO_PIL_STASH", True), ("CE_TA_LP1", True), ("CE_TA_FULL", False),
                                          ("CE_FORCE_GENERIC", False), ("CE_FORCE_WIDE", False)])
def test_alternative_kernel_paths_agree(tmp_path, default_results, knob, bitwise):
    alt = _run(tmp_path, knob, also=("CE_NO_NARROW",))
    for c in CASES:
        n = c["name"]
        ch0, ch1 = default_results[n + "/ch"], alt[n + "/ch"]
        if bitwise or knob == "CE_TA_FULL":   # the TA transform never touches the grid
            assert np.array_equal(ch0, ch1), f"{knob}: {n} channel estimate differs"
        else:
            assert np.abs(ch0 - ch1).max() <= 2e-6 * np.abs(ch0).max(), f"{knob}: {n}"
        assert np.array_equal(default_results[n + "/ta"], alt[n + "/ta"]), f"{knob}: {n} time alignment"
        for k in ("noise", "rsrp", "epre", "cfo"):
            a, b = default_results[n + "/" + k], alt[n + "/" + k]
            if bitwise:
                assert np.array_equal(a, b), f"{knob}: {n} {k}"
            else:
                assert np.allclose(a, b, rtol=2e-6, atol=1e-9 if k != "cfo" else 1e-3), f"{knob}: {n} {k}: {a} vs {b}"
