"""Per-stage parity: the fused kernel never materialises process_hop's intermediates, so a
diagnostic launch (``ce_estimate_batch_stages``, include/ce_hip.h) dumps them -- the pilot-RE channel estimate after
LS + DM-RS average + CDM de-spread (S5/S6, T:593-628) and after frequency smoothing (S7, T:633-668), each hop's CFO
(S4, T:426) and its time-alignment arg-max bin (S8, T:686-696) -- and each is compared with the same stage of the CPU
oracle on the reference's own fixture inputs.  The six ordinary outputs of the diagnostic launch must be bit-identical
to the ordinary launch's."""
import numpy as np
import pytest
import torch

from conftest import load_fixture

import ce_oracle as O
from srsran_ce_pytorch_amd import estimator as E

pytestmark = pytest.mark.gpu

FIXTURES = ["pusch273_filter_L1", "pusch273_none_L1", "pusch273_mean_L1", "cfg1_25prb_1dmrs_none", "hop2_2dmrs_each",
            "case4like_fullslot_hops", "layers2_6prb", "layers4_6prb", "layers4_273", "type2_layers3", "prb1_filter",
            "dmrs4_20prb", "hop2_136prb_273", "dmrs3_273", "hops_partial_overlap_L2"]
TOL_P = 2e-6      # float32 pipelines on both sides (measured ~3e-7 of max|P|)


@pytest.mark.parametrize("name", FIXTURES)
def test_stages_match_the_oracle(name):
    fx = load_fixture(name)
    dev = torch.device("cuda:0")
    n_items, n_hops, L = fx.grids.shape[0], len(fx.case["hops"]), fx.case["n_layers"]
    g = torch.as_tensor(fx.grids, device=dev)[None].permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    pil = torch.as_tensor(fx.pilots, device=dev)
    plan = E.make_plan(fx.hop1, fx.hop2, fx.config, fx.beta, L, fx.case["n_prb_grid"], fx.case["n_sym"], dev)
    plain = E.estimate_with_plan(plan, g, pil)
    out, st_p, st_s = E.estimate_stages(plan, g, pil, n_hops)
    torch.cuda.synchronize()
    for a, b in zip(plain, out):                                  # the dump changes nothing
        assert torch.equal(a, b) or (torch.isnan(a).all() and torch.isnan(b).all())
    st_p, st_s = st_p[0].cpu().numpy(), st_s[0].cpu().numpy()
    for it in range(n_items):
        stages = []
        O.srs_channel_estimator(fx.grids[it], fx.pilots, fx.beta, fx.hop1, fx.hop2, fx.config, stages=stages)
        assert len(stages) == n_hops
        for h, st in enumerate(stages):
            for k, key in enumerate(("p_ls", "p_smooth")):
                want = st[key].T                                    # oracle: (n_re, L) -> [L][n_re]
                got = st_p[it, k, h]
                err = np.abs(got - want).max() / np.abs(want).max()
                assert err <= TOL_P, f"{name}[{it}] hop {h} {key}: {err:.2e}"
            if st["cfo_hop"] is None:
                assert np.isnan(st_s[it, h, 0])
            else:
                assert abs(st_s[it, h, 0] - st["cfo_hop"]) <= 2e-6 * max(abs(st["cfo_hop"]), 1e-3), f"{name}[{it}] hop {h} cfo"
            assert st_s[it, h, 1] == float(st["ta_bin"]), f"{name}[{it}] hop {h} TA bin {st_s[it, h, 1]} vs {st['ta_bin']}"
