"""CPU-side checks of bench.py: the workload table, the CPU-baseline leg (the oracle timed on a tiny sample) and the
line's schema helpers -- so a broken baseline leg cannot take the driver's bench run down on the GPU box."""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
sys.modules["bench"] = bench          # the worker pool pickles bench._cpu_worker by module name
spec.loader.exec_module(bench)


def test_workloads_name_baseline_configs():
    assert bench.WORKLOADS["pusch273_4rx_filter"] == dict(smoothing="filter", ports=4, slots=8192)      # configs[2], the default
    assert bench.WORKLOADS["pusch273_1rx_none"] == dict(smoothing="none", ports=1, slots=1024)          # configs[1]
    assert bench.WORKLOADS["pusch273_4rx_cnn"]["interp"] == "cnn" and bench.WORKLOADS["pusch273_4rx_denoise"]["denoise"]
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.MFMA_F16_PEAK_TFLOPS == 2500.0
    assert bench.DENOISE_FLOP_PER_PIXEL == 5760


def test_cpu_baseline_leg_runs_on_a_small_sample():
    from srsran_ce_pytorch_amd import synth as S
    case = S.case_spec("tiny", 25, [S.hop_spec([2, 11], 0, 25)], seed=3)
    for flavour, denoise in (("tensorized", False), ("tensorized", True), ("baseline_loop", False), ("cnn", False)):
        items, seconds = bench._cpu_worker((case, 2, 1, 7, flavour, denoise))
        assert items == 2 and seconds > 0
    out = bench.cpu_baseline(case, 2, target_core_seconds=0.2)
    assert out["unit"] == "slots/s" and out["kind"] == "port" and out["value"] > 0 and 1 <= out["cores"] <= 16
    # `value` is the loop-style ce_rule_baseline port (the baseline north_star names); the tensorized port sits next to it
    # (no ordering between the two is asserted: 0.2 core-seconds on a busy host say nothing about speed)
    assert out["flavour"].startswith("ce_rule_baseline") and out["tensorized_value"] > 0 and out["config0_ms"] > 0
    json.dumps(out)
    cnn = bench.cpu_baseline(case, 2, target_core_seconds=0.2, interp="cnn")
    assert cnn["flavour"] == "ce_dl_cnn" and cnn["value"] > 0 and "tensorized_value" not in cnn
