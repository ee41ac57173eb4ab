#!/usr/bin/env python3
"""Build-container-only dev check (needs /root/reference; the GPU box has none): the REFERENCE's own harness
(scripts/validation/validate_all.py: its header parser, its .dat readers, its pilot-order search) running the
REFERENCE's own estimator on CPU over the synthetic srsRAN-style vector sets that tests/test_vector_header.py writes.
Errors ~1e-7 mean this build's file formats, header layout, pilot orders and two-hop convention are the harness's.

    python tools/crosscheck_harness.py"""
import pathlib, sys, tempfile
ROOT = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path("/root/reference")
if not REF.exists():
    sys.exit("needs the reference checkout at /root/reference (build container only)")
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests"), str(REF / "src"), str(REF / "scripts" / "validation")]
import test_vector_header as TV          # noqa: E402
import validate_all as VA                # noqa: E402  (imports ce_rule_tensorized from the reference's src/)

with tempfile.TemporaryDirectory() as d:
    tmp = pathlib.Path(d)
    header, truth = TV._build(tmp)
    VA.HEADER_PATH, VA.DATA_DIR, VA.DEBUG_CASES = tmp / "port_channel_estimator_test_data.h", tmp, set()
    worst = 0.0
    for case in VA.parse_header():
        if truth[case.idx]["spec"]["grid"] != 52:
            print(f"set {case.idx}: skipped (the harness hard-codes 52-PRB masks, validate_all.py:171)")
            continue
        mx, rms = VA.run_case(case)
        worst = max(worst, mx)
        print(f"set {case.idx}: {len(case.hops)} hop(s) as parsed by the harness, max {mx:.2e} rms {rms:.2e}")
    print("worst:", worst)
    sys.exit(0 if worst < 1e-5 else 1)
