"""N>1 path on CPU: world_size-2 gloo processes exercise the slot sharding and the max-over-ranks timing
reduction bench.py uses (the data path itself has no collective to test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from srsran_ce_pytorch_amd.sharding import aggregate_slots_per_second, max_over_ranks, shard_slots


def test_shard_slots_partition():
    for n, w in [(8192, 8), (65536, 8), (10, 3), (3, 8), (0, 2)]:
        ranges = [shard_slots(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_slots(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, stop = shard_slots(1001, rank, world)
        # each rank "processes" its shard; totals are checked with an all_reduce the bench never needs
        t = torch.tensor([float(stop - start)], dtype=torch.float64)
        dist.all_reduce(t)
        elapsed, kernel_ms = max_over_ranks([0.5 + rank, 2.0 - rank])
        dist.barrier()
        q.put((rank, float(t[0]), elapsed, kernel_ms, aggregate_slots_per_second(500, 4, elapsed, world)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_timing_reduction():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, elapsed, kernel_ms, agg in res:
        assert total == 1001.0                      # shards cover the batch exactly once
        assert elapsed == 1.5 and kernel_ms == 2.0  # MAX over ranks of (0.5, 1.5) and (2.0, 1.0)
        assert agg == pytest.approx(2 * 500 * 4 / 1.5)
