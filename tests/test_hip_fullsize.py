"""GPU tests at BASELINE.json's full sizes (273 PRB, 4 Rx, 8192 slots) through size-independent
properties, plus the edge cases of the boundary (empty batch, non-14-symbol grid, maximum grid width)."""
import numpy as np
import pytest
import torch

from conftest import check_outputs

import ce_oracle as O
from srsran_ce_pytorch_amd import estimator as E, synth as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def full_batch():
    case = S.bench_case("filter", 1, seed=4321)
    h1, h2, cfg = S.numpy_hops(case)
    dev = torch.device(DEV)
    plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
    rx, pil = S.torch_inputs(case, 8192, 4, dev, seed=4321)
    ch = torch.full((8192, 4, 3276, 14, 1), float("nan"), dtype=torch.complex64, device=dev)
    sc = torch.full((5, 8192, 4), float("nan"), dtype=torch.float64, device=dev)
    out = E.estimate_with_plan(plan, rx, pil, (ch, sc[0], sc[1], sc[2], sc[3], sc[4]))
    torch.cuda.synchronize()
    return dict(case=case, hops=(h1, h2, cfg), plan=plan, rx=rx, pil=pil, out=out)


def test_full_size_every_output_written_and_sane(full_batch):
    out = full_batch["out"]
    assert not bool(torch.isnan(torch.view_as_real(out[0])).any())
    for t in out[1:]:
        assert bool(torch.isfinite(t).all())
    # generated CFO is 125..375 Hz either sign, TA is the 200 ns bulk delay quantised to 1/(4096*scs)
    assert bool(((out[5].abs() > 100) & (out[5].abs() < 400)).all())
    assert bool(((out[4] > 1.5e-7) & (out[4] < 2.6e-7)).all())
    assert bool((out[1] > 0).all()) and bool((out[2] > out[1]).all())     # noise > 0, RSRP > noise


def test_full_size_items_independent_of_batch(full_batch):
    """Slots are independent work items: a sub-batch must reproduce the big batch bit for bit."""
    fb = full_batch
    for sl in (slice(0, 3), slice(4095, 4098), slice(8189, 8192)):
        sub = E.estimate_with_plan(fb["plan"], fb["rx"][sl], fb["pil"][sl])
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(sub[0]), torch.view_as_real(fb["out"][0][sl]))
        for a, b in zip(sub[1:], fb["out"][1:]):
            assert torch.equal(a, b[sl])


def test_full_size_power_of_two_scaling_is_exact(full_batch):
    """Linearity: doubling the received grid doubles h and quadruples the powers, exactly in binary fp."""
    fb = full_batch
    sl = slice(1000, 1512)
    base = [t[sl] for t in fb["out"]]
    got = E.estimate_with_plan(fb["plan"], fb["rx"][sl] * 2.0, fb["pil"][sl])
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(got[0]), torch.view_as_real(base[0]) * 2.0)
    for i in (1, 2, 3):
        assert torch.allclose(got[i], base[i] * 4.0, rtol=1e-12, atol=0)
    assert torch.equal(got[4], base[4]) and torch.allclose(got[5], base[5], rtol=1e-9, atol=1e-9)


def test_full_size_common_phase_rotation(full_batch):
    """A common phase on the received grid rotates h and leaves every scalar alone."""
    fb = full_batch
    sl = slice(2000, 2256)
    rot = complex(np.cos(0.7), np.sin(0.7))
    got = E.estimate_with_plan(fb["plan"], fb["rx"][sl] * rot, fb["pil"][sl])
    torch.cuda.synchronize()
    ref = fb["out"][0][sl] * rot
    assert float((got[0] - ref).abs().max() / ref.abs().max()) < 5e-6
    for i in (1, 2, 3, 5):
        assert torch.allclose(got[i], fb["out"][i][sl], rtol=2e-5, atol=1e-9)
    assert torch.equal(got[4], fb["out"][4][sl])


def test_full_size_spot_check_against_oracle(full_batch):
    fb = full_batch
    h1, h2, cfg = fb["hops"]
    rng = np.random.default_rng(5)
    for slot, port in zip(rng.integers(0, 8192, 6), rng.integers(0, 4, 6)):
        rg = fb["rx"][slot, port].cpu().numpy()
        ref = O.srs_channel_estimator(rg, fb["pil"][slot].cpu().numpy(), fb["case"]["beta"], h1, h2, cfg)
        got = [float(fb["out"][i][slot, port]) for i in range(1, 6)]
        check_outputs(fb["out"][0][slot, port].cpu().numpy(), got, ref[0], list(ref[1:]), 2e-5, 2e-5, f"slot {slot} port {port}")


def test_empty_batch_and_single_item():
    case = S.case_spec("edge", 52, [S.hop_spec([2, 11], 4, 9)], seed=77)
    b = S.build_case(case, 1)
    dev = torch.device(DEV)
    pil = torch.as_tensor(b.pilots, device=dev)
    out = E.estimate(torch.empty((0, 2, 624, 14), dtype=torch.complex64, device=dev), pil, b.beta, b.hop1, b.hop2, b.config)
    assert tuple(out[0].shape) == (0, 2, 624, 14, 1) and out[1].numel() == 0
    one = E.estimate(torch.as_tensor(b.grids, device=dev)[None], pil, b.beta, b.hop1, b.hop2, b.config)
    ref = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)
    check_outputs(one[0][0, 0].cpu().numpy(), [float(one[i][0, 0]) for i in range(1, 6)], ref[0], list(ref[1:]), 2e-5, 2e-5, "single")


def test_twelve_symbol_grid_without_cfo():
    """n_sym != 14 is legal when no hop has two DM-RS symbols (the CFO ramp T:928-929 is then skipped):
    exercises the generic element-wise writer."""
    case = S.case_spec("sym12", 24, [S.hop_spec([3], 2, 20, 0, 12)], n_sym=12, smoothing="filter", seed=78)
    b = S.build_case(case, 2)
    dev = torch.device(DEV)
    out = E.estimate(torch.as_tensor(b.grids, device=dev)[None], torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config)
    assert out[5].numel() == 0
    for it in range(2):
        ref = O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config)
        got = [float(out[i][0, it]) for i in range(1, 5)] + [np.nan]
        check_outputs(out[0][0, it].cpu().numpy(), got, ref[0], [ref[1], ref[2], ref[3], ref[4], np.nan], 2e-5, 2e-5, f"sym12[{it}]")


def test_maximum_grid_width_and_unsupported_sizes():
    """341 PRB = 4092 subcarriers is the widest grid the 4096-point TA transform admits (T:679)."""
    case = S.case_spec("wide", 341, [S.hop_spec([2, 11], 0, 341)], smoothing="none", seed=79)
    b = S.build_case(case, 1)
    dev = torch.device(DEV)
    out = E.estimate(torch.as_tensor(b.grids, device=dev)[None], torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config)
    ref = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)
    check_outputs(out[0][0, 0].cpu().numpy(), [float(out[i][0, 0]) for i in range(1, 6)], ref[0], list(ref[1:]), 2e-5, 2e-5, "wide")
    h1, h2, cfg = S.numpy_hops(S.case_spec("too_wide", 400, [S.hop_spec([2, 11], 0, 400)]))
    with pytest.raises(NotImplementedError):            # 4800 subcarriers: the reference would truncate its IFFT input
        E.make_plan(h1, h2, cfg, 1.0, 1, 400, 14, dev)


@pytest.mark.parametrize("two_hops", [False, True])
def test_scattered_mask_at_the_top_of_the_widest_grid(two_hops):
    """A narrow scattered PRB mask on the last PRBs of a 341-PRB grid: the collapsed first pass of the TA transform probes
    subcarriers shift .. shift + 511 of the hop's subcarrier -> pilot table, which must stay inside its 4096 entries (with
    two hops the entries behind them are the second hop's, whose pilots here sit exactly where an over-run would look)."""
    top = [330, 332, 335, 337, 339, 340]
    hops = [S.hop_spec([2, 5] if two_hops else [2, 11], 335, 6, 0, 7 if two_hops else 14, mask_prbs=top)]
    if two_hops:
        hops.append(S.hop_spec([8, 12], 0, 6, 7, 7, mask_prbs=[0, 1, 3, 5, 20, 21]))
    case = S.case_spec("top_scatter", 341, hops, smoothing="filter", seed=83)
    b = S.build_case(case, 2)
    dev = torch.device(DEV)
    out = E.estimate(torch.as_tensor(b.grids, device=dev)[None], torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config)
    for it in range(2):
        ref = O.srs_channel_estimator(b.grids[it], b.pilots, b.beta, b.hop1, b.hop2, b.config)
        check_outputs(out[0][0, it].cpu().numpy(), [float(out[i][0, it]) for i in range(1, 6)], ref[0], list(ref[1:]), 2e-5, 2e-5, f"top_scatter[{it}]")


def test_large_plan_survives_a_later_small_plan_of_the_same_kernel():
    """Plans of different LDS sizes share a kernel instantiation; the dynamic-LDS limit of the instantiation must
    only ever grow, or relaunching the cached large plan after a small one was created would fail (or clip)."""
    dev = torch.device("cuda:0")
    BOTH = [S.TYPE1_CDM0, S.TYPE1_CDM1]
    big = S.case_spec("l4_big", 273, [S.hop_spec([2, 11], 0, 273, re_masks=BOTH)], n_layers=4, seed=3)
    small = S.case_spec("l4_small", 52, [S.hop_spec([2, 11], 5, 6, re_masks=BOTH)], n_layers=4, seed=4)
    outs = []
    for case in (big, small, big):
        b = S.build_case(case, 1)
        h1, h2, cfg = S.numpy_hops(case)
        plan = E.make_plan(h1, h2, cfg, case["beta"], 4, case["n_prb_grid"], 14, dev)
        rg = torch.as_tensor(b.grids, device=dev)[None]
        out = E.estimate_with_plan(plan, rg, torch.as_tensor(b.pilots, device=dev))
        torch.cuda.synchronize()
        outs.append((plan.lds_bytes, out[0].clone(), b))
    assert outs[0][0] > outs[1][0]
    assert torch.equal(outs[0][1], outs[2][1])
    b = outs[0][2]
    ref = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)
    assert np.abs(outs[2][1][0, 0].cpu().numpy() - ref[0]).max() <= 2e-5 * np.abs(ref[0]).max()


def test_large_none_launch_with_the_big_lds_request_equals_small_launches():
    """From 8192 work items on, launches of the wide none / mean kernel ask for enough dynamic LDS to leave room for two
    workgroups per CU only (ce_api.hip: lds_big).  Placement only: a 2304 x 4 launch must reproduce, bit for bit, the same
    slots estimated 512 at a time (below the threshold)."""
    case = S.bench_case("none", 1, seed=97)
    h1, h2, cfg = S.numpy_hops(case)
    dev = torch.device(DEV)
    plan = E.make_plan(h1, h2, cfg, case["beta"], 1, 273, 14, dev)
    rx, pil = S.torch_inputs(case, 2304, 4, dev, seed=97)
    big = E.estimate_with_plan(plan, rx, pil)
    torch.cuda.synchronize()
    for s0 in (0, 1024, 1792):
        sl = slice(s0, s0 + 512)
        sub = E.estimate_with_plan(plan, rx[sl], pil[sl])
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(sub[0]), torch.view_as_real(big[0][sl]))
        for a, b in zip(sub[1:], big[1:]):
            assert torch.equal(a, b[sl])


@pytest.mark.parametrize("case", [
    S.case_spec("nrw_case4like", 52, [S.hop_spec([0, 4], 3, 3, 0, 14), S.hop_spec([8, 12], 28, 3, 0, 14)], scs=15e3, seed=71),
    S.case_spec("nrw_L2_2hop_12prb", 52, [S.hop_spec([1, 5], 3, 12, 0, 7), S.hop_spec([8, 12], 30, 12, 7, 7)], n_layers=2, seed=72),
], ids=lambda c: c["name"])
def test_wave_per_item_kernel_at_full_batch_size(case):
    """The wave-per-item kernel (csrc/ce_narrow_kernel.h) at the bench's batch size -- 8192 slots x 4 ports = 8192 workgroups of
    four items: every output written and finite, items independent of the batch (a large launch equals small launches bit for
    bit, also across the last, partly filled workgroup of an odd-sized batch), and a spot check against the oracle."""
    dev = torch.device(DEV)
    h1, h2, cfg = S.numpy_hops(case)
    L = case["n_layers"]
    assert E.derive_host(h1, h2, cfg, case["beta"], L, 52, 14).narrow == 1
    plan = E.make_plan(h1, h2, cfg, case["beta"], L, 52, 14, dev)
    rx, pil = S.torch_inputs(case, 8192, 4, dev, seed=73)
    ch = torch.full((8192, 4, 624, 14, L), float("nan"), dtype=torch.complex64, device=dev)
    sc = torch.full((5, 8192, 4), float("nan"), dtype=torch.float64, device=dev)
    out = E.estimate_with_plan(plan, rx, pil, (ch, sc[0], sc[1], sc[2], sc[3], sc[4]))
    torch.cuda.synchronize()
    assert not bool(torch.isnan(torch.view_as_real(out[0])).any())
    assert all(bool(torch.isfinite(t).all()) for t in out[1:])
    for a, b in ((0, 3), (4000, 4001), (8189, 8192)):          # 3 slots x 4 ports = 12 items, 1 slot, the batch's last 3 slots
        part = E.estimate_with_plan(plan, rx[a:b], pil[a:b])
        for x, y in zip(out, part):
            assert torch.equal(x[a:b], y), f"slots {a}:{b} differ between the large launch and a small one"
    odd = E.estimate_with_plan(plan, rx[:5, :3], pil[:5])      # 15 items: the last workgroup carries three live waves
    for x, y in zip(out, odd):
        assert torch.equal(x[:5, :3], y)
    b = S.build_case(case, 1)                                   # same geometry, numpy inputs: the oracle as the checker
    got = E.estimate(torch.as_tensor(b.grids, device=dev)[None], torch.as_tensor(b.pilots, device=dev), b.beta, b.hop1, b.hop2, b.config)
    ref = O.srs_channel_estimator(b.grids[0], b.pilots, b.beta, b.hop1, b.hop2, b.config)
    check_outputs(got[0][0, 0].cpu().numpy(), [float(got[i][0, 0]) for i in range(1, 6)], ref[0], list(ref[1:]), 2e-5, 2e-5, case["name"])
