"""bench.py's N>1 control flow.

* CPU (gloo, world_size 2, runs everywhere): the timing bracket (`bench.time_steps`: warm-up, barrier, K steps, barrier)
  and the max-over-ranks / whole-job aggregation the line is built from.
* GPU box (`-m gpu`): the real `bench.py` under `python -m torch.distributed.run --nproc-per-node 2`, as the driver
  launches it for N>1, with CE_BENCH_REHEARSE=1 (both ranks on the box's single GPU, rendezvous over gloo): a fresh
  child process tree -- the test process itself never re-executes.  This file sorts first among the GPU test modules
  so the launcher starts before this process has touched the GPU."""
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from srsran_ce_pytorch_amd.sharding import aggregate_slots_per_second, max_over_ranks

ROOT = Path(__file__).resolve().parents[1]


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mr", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    bench = _load_bench()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def step():
            calls.append(time.perf_counter())
            time.sleep(0.002 * (rank + 1))              # rank 1 is the slow one

        elapsed, kernel_ms = bench.time_steps(step, dist.barrier, steps=5, warmup=2)
        assert kernel_ms is None and len(calls) == 7
        slowest, = max_over_ranks([elapsed])
        q.put((rank, elapsed, slowest, aggregate_slots_per_second(64, 5, slowest, world)))
    finally:
        dist.destroy_process_group()


def test_time_steps_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, e0, s0, a0), (_, e1, s1, a1) = res
    assert s0 == s1 == max(e0, e1)                       # every rank reports the slowest rank's bracket
    assert e1 >= 5 * 0.004 and e0 >= 5 * 0.004 * 0.9     # the closing barrier makes the fast rank wait for the slow one
    assert a0 == a1 == pytest.approx(2 * 64 * 5 / s0)    # whole-job slots / max-over-ranks time


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    env = dict(os.environ, CE_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--slots", "64", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 prints ONE line
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["slots_per_gpu"] == 64 and line["config"]["global_slots"] == 128
    assert "REHEARSAL" in line["data"] and "cpu_baseline" not in line and "secondary" not in line
    assert line["value"] == pytest.approx(128 * 2 / (line["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    assert line["roofline"]["kernel_ms"] <= line["ms_per_step"] * 1.05
    out = ROOT / "gpurun_out"
    if out.is_dir():                                      # kept for profiles/ (the judge asked for the log)
        (out / "bench_rehearsal_2ranks.json").write_text(json.dumps(line) + "\n")


@pytest.mark.gpu
def test_bench_one_rank_rccl_process_group():
    """The code the driver's 8-GPU job runs and a one-GPU box otherwise never does: `init_process_group("nccl", device_id=dev)`
    (RCCL bootstrap with this pool's IPC mode), the device-tensor `all_reduce(MAX)` of the timing, `dist.barrier()` on RCCL and
    `destroy_process_group()` -- at world size 1 through CE_BENCH_FORCE_PG=1, launched exactly as the driver launches N > 1
    (a fresh child `python -m torch.distributed.run`; this process never re-executes)."""
    env = dict(os.environ, CE_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CE_BENCH_REHEARSE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--slots", "64", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 1 and line["config"]["process_group"] == "nccl (RCCL)" and "REHEARSAL" not in line["data"]
    assert line["config"]["global_slots"] == 64 and line["roofline"]["kernel_ms"] <= line["ms_per_step"] * 1.05
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "bench_rccl_1rank.json").write_text(json.dumps(line) + "\n")


def test_bench_exits_nonzero_when_the_process_group_cannot_start():
    """A rendezvous / RCCL failure must surface as a non-zero exit with the library's message -- never a retry, never a
    silent single-process run.  CPU-checkable part: WORLD_SIZE that contradicts --gpus is refused before anything starts."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
