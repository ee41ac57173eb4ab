"""`sharding.estimate_sharded`: one host thread, one plan + one launch stream per shard.  A one-GPU box
exercises it with `devices=[0, 0]` -- two shards, two streams, one device -- and with a single shard; results must be
bit-identical to the unsharded call (placement never changes results).  Shard arithmetic and argument checks run on CPU."""
import numpy as np
import pytest
import torch

from srsran_ce_pytorch_amd import estimator as E, sharding as SH, synth as S


def _case():
    return S.case_spec("shard", 52, [S.hop_spec([2, 11], 6, 12)], seed=611)


def test_estimate_sharded_argument_checks():
    rx = torch.zeros((2, 1, 624, 14), dtype=torch.complex64)
    pil = torch.zeros((72, 2, 1), dtype=torch.complex64)
    h1, h2, cfg = S.numpy_hops(_case())
    with pytest.raises(ValueError, match="one entry per shard"):
        SH.estimate_sharded([rx, rx], [pil], 1.0, h1, h2, cfg)
    with pytest.raises(ValueError, match="2 shards but 1 devices"):
        SH.estimate_sharded([rx, rx], [pil, pil], 1.0, h1, h2, cfg, devices=[0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SH.estimate_sharded([rx], [pil], 1.0, h1, h2, cfg)           # CPU tensors: refused, never estimated on the host


@pytest.mark.gpu
@pytest.mark.parametrize("per_slot_pilots", [False, True])
def test_estimate_sharded_two_streams_on_one_device_is_bit_identical(per_slot_pilots):
    dev = torch.device("cuda:0")
    case = _case()
    n_slots, n_ports = 37, 2                                          # odd: shards of 19 and 18 slots
    rx, pil = S.torch_inputs(case, n_slots, n_ports, dev, seed=5)
    if not per_slot_pilots:
        pil = pil[0]
    h1, h2, cfg = S.numpy_hops(case)
    whole = E.estimate(rx, pil, case["beta"], h1, h2, cfg)
    rx_s, pil_s = SH.split_slots(rx, pil, [0, 0])
    assert [t.shape[0] for t in rx_s] == [19, 18] and rx_s[0].stride() == rx[:19].stride()
    res = SH.estimate_sharded(rx_s, pil_s, case["beta"], h1, h2, cfg, devices=[0, 0])
    assert len(res) == 2 and len({id(SH._shard_stream(dev, k)) for k in range(2)}) == 2
    # consumed on the current stream without a host synchronisation: the current stream waits for both launch streams
    for k in range(6):
        got = torch.cat([r[k] for r in res], dim=0)
        assert torch.equal(got, whole[k]), f"output {k} differs between sharded and unsharded"
    # caller-provided outputs are overwritten in place; a single shard equals the whole batch
    outs = [tuple(torch.full_like(t, float("nan")) for t in r) for r in res]
    res2 = SH.estimate_sharded(rx_s, pil_s, case["beta"], h1, h2, cfg, devices=[dev, dev], outs=outs)
    torch.cuda.synchronize()
    for r2, o, r in zip(res2, outs, res):
        assert all(a.data_ptr() == b.data_ptr() and torch.equal(a, c) for a, b, c in zip(r2, o, r))
    one = SH.estimate_sharded([rx], [pil], case["beta"], h1, h2, cfg)
    assert all(torch.equal(a, b) for a, b in zip(one[0][:5], whole[:5]))
