#!/usr/bin/env python3
"""gpurun_out/<p>_bench_{trace,fetch,write,sq}/ + <p>_bench_same_lease.log + <p>_rwmix.log (one gpurun lease,
tools/gpu_profile_round.sh) -> gpurun_out/distilled/<tag>_{summary.json,kernel_stats.csv,bench_line.json} (copied to profiles/).

    python tools/distill_round.py <tag, e.g. round3> <prefix of the gpurun_out directories, e.g. r3> <git head the lease ran at>

The kernel's register / LDS / occupancy figures come from the COMPILER (hipcc -Rpass-analysis=kernel-resource-usage of the
headline kernel's translation unit with that unit's own flags, as tools/kernel_resources.py does) and from the plan
(dynamic LDS bytes in the bench line) -- rocprofv3's VGPR_Count / LDS_Block_Size columns are not what the kernel holds
(round 2's summary carried 96 VGPRs / 0 B for a kernel with 191 / 36 832) -- and the residency the SQ counters measure is
checked against the bound those figures give."""
import csv, glob, hashlib, json, os, re, statistics as st, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
tag = sys.argv[1] if len(sys.argv) > 1 else "round3"
PFX = sys.argv[2] if len(sys.argv) > 2 else "r3"
HEAD = sys.argv[3] if len(sys.argv) > 3 else None
OUT = "gpurun_out/distilled"   # gpurun merges only gpurun_out/ back; copy the files into profiles/ afterwards
os.makedirs(OUT, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
ks = list(csv.DictReader(open(newest(f"gpurun_out/{PFX}_bench_trace/*/*_kernel_stats.csv"))))

with open(f"{OUT}/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(ks[0].keys())
    for r in ks[:12]:
        r = dict(r); r["Name"] = r["Name"][:160]; w.writerow(r.values())
ce = [r for r in ks if "ce_estimate" in r["Name"]][0]
out = {"lease": "one gpurun call: bench.py, tools/micro/rwmix.hip, rocprofv3 trace and PMC passes back to back on the same box",
       "command": "rocprofv3 --kernel-trace --stats --kernel-include-regex ce_estimate --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary",
       "kernel": ce["Name"][:90], "calls": int(ce["Calls"]), "avg_ns": float(ce["AverageNs"]), "min_ns": int(ce["MinNs"]), "max_ns": int(ce["MaxNs"])}


def counter(d, name):
    rows = [r for r in csv.DictReader(open(newest(f"gpurun_out/{d}/*/*_counter_collection.csv"))) if "ce_estimate" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return rows, [float(r["Counter_Value"]) for r in rows]


for name, d in (("FETCH_SIZE", f"{PFX}_bench_fetch"), ("WRITE_SIZE", f"{PFX}_bench_write")):
    rows, vals = counter(d, name)
    out[name + "_KB_per_launch_median"] = st.median(vals); out[name + "_launches"] = len(vals)
    out["grid_threads"] = rows[0]["Grid_Size"]; out["workgroup"] = rows[0]["Workgroup_Size"]
fetch_b = out["FETCH_SIZE_KB_per_launch_median"] * 1024 * 2   # gfx950: FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM)
write_b = out["WRITE_SIZE_KB_per_launch_median"] * 1024
alg = 13096452096


def source_sha():
    """What bench.py compares against to decide whether this profile still describes the kernel it runs (roofline.traffic)."""
    h = hashlib.sha256()
    for f in ("ce_estimate_kernel.h", "ce_plan.h", "ce_api.hip", "ce_inst.inc"):
        h.update((ROOT / "srsran_ce_pytorch_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


out["head"] = HEAD
out["kernel_source_sha16"] = source_sha()
out.update(hbm_read_bytes_per_launch_corrected=fetch_b, hbm_write_bytes_per_launch=write_b, hbm_traffic_bytes_per_launch=fetch_b + write_b,
           algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(fetch_b + write_b) / alg,
           achieved_GBps_algorithmic=alg / (out["avg_ns"] * 1e-9) / 1e9, achieved_GBps_traffic=(fetch_b + write_b) / (out["avg_ns"] * 1e-9) / 1e9,
           pmc_commands=["rocprofv3 --kernel-include-regex ce_estimate --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary",
                         "rocprofv3 --kernel-include-regex ce_estimate --pmc WRITE_SIZE ... (same)",
                         "rocprofv3 --kernel-include-regex ce_estimate --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE ... (same)"],
           workload="pusch273_4rx_filter (8192 slots x 4 ports, 273 PRB)")
sq = {}
for c in ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]:
    try:
        sq[c] = st.median(counter(f"{PFX}_bench_sq", c)[1])
    except Exception as e:
        sq[c] = None
if sq.get("SQ_WAVE_CYCLES"):
    wc = sq["SQ_WAVE_CYCLES"]
    sq.update(wave_parked_frac=sq["SQ_WAIT_ANY"] / wc, wave_issue_stall_frac=sq["SQ_WAIT_INST_ANY"] / wc, wave_issuing_frac=sq["SQ_ACTIVE_INST_ANY"] / wc,
              wave_lifetime_shader_cycles=4.0 * wc / sq["SQ_WAVES"],
              note="SQ_*_CYCLES count quad-cycles; WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (MI355X_MICROARCH.md, rocprofv3 PMC slots)")
    if sq.get("GRBM_GUI_ACTIVE"):
        sq["resident_waves_per_cu"] = 4.0 * wc / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
out["sq"] = sq
# same-lease: plain bench line and the access-pattern micro-benchmark
line = [l for l in open(f"gpurun_out/{PFX}_bench_same_lease.log") if l.startswith("{")][-1]
open(f"{OUT}/{tag}_bench_line.json", "w").write(line)
bl = json.loads(line)
out["same_lease_bench"] = {"ms_per_step": bl["ms_per_step"], "kernel_ms": bl["roofline"]["kernel_ms"], "roofline_frac": bl["roofline"]["frac"],
                           "secondary": [(s["workload"], s.get("rx_layout", ""), round(s["ms_per_step"], 4), round(s["roofline"]["frac"], 4)) for s in bl.get("secondary", [])],
                           "cpu_baseline": {k: bl["cpu_baseline"][k] for k in bl.get("cpu_baseline", {}) if k != "sample"}}
rw = {}
for l in open(f"gpurun_out/{PFX}_rwmix.log"):
    m = re.match(r"\s*(.*?)\s*:\s*([\d.]+) ms\s+([\d.]+) GB/s(.*)", l)
    if m:
        rw[m.group(1).strip()] = {"ms": float(m.group(2)), "GBps": float(m.group(3)), "of": m.group(4).strip()}
out["same_lease_rwmix"] = rw
# resources of the headline kernel from the compiler's remarks (its own unit, its own flags) + the plan's dynamic LDS
try:
    from kernel_resources import resources, short
    from srsran_ce_pytorch_amd._lib import EXTRA_FLAGS
    unit = "ce_inst_reg_h1_f1w.hip"
    rows = [r for r in resources(ROOT / "srsran_ce_pytorch_amd" / "csrc" / unit, EXTRA_FLAGS.get(unit, [])) if short(r["name"]) == "<L1,NH1,ND2,KPT7,F1>"]
    r = rows[0]
    vg, lds = int(r["VGPRs"]), int(bl["roofline"].get("lds_bytes_per_workgroup") or 0)
    alloc = (vg + 7) // 8 * 8                                     # VGPRs are granted in blocks of 8 (512 per SIMD lane)
    waves_simd = min(8, 512 // alloc)
    wg_regs = waves_simd * 4 // 4                                 # 4 SIMDs, 4 waves per workgroup: one wave of every workgroup per SIMD
    wg_lds = (160 * 1024) // ((lds + 2047) // 2048 * 2048) if lds else None
    wg = min(wg_regs, wg_lds) if wg_lds else wg_regs
    out["kernel_resources"] = {"source": f"hipcc -Rpass-analysis=kernel-resource-usage, {unit} with {' '.join(EXTRA_FLAGS.get(unit, []))}",
                               "vgprs": vg, "agprs": int(r.get("AGPRs", 0)), "sgprs": int(r.get("TotalSGPRs", 0)), "vgpr_spill": int(r.get("VGPRs Spill", 0)),
                               "scratch_bytes_per_lane": int(r.get("ScratchSize [bytes/lane]", 0)), "lds_bytes_per_workgroup": lds,
                               "workgroups_per_cu_by_registers": wg_regs, "workgroups_per_cu_by_lds": wg_lds, "max_resident_waves_per_cu": 4 * wg}
    if sq.get("resident_waves_per_cu"):
        assert sq["resident_waves_per_cu"] <= 4 * wg * 1.05, f"SQ residency {sq['resident_waves_per_cu']:.1f} waves/CU exceeds the bound {4 * wg} the compiler's figures give"
        out["kernel_resources"]["measured_resident_waves_per_cu"] = sq["resident_waves_per_cu"]
except Exception as e:                                            # (no hipcc on the box, unit renamed ...): say so instead of guessing
    out["kernel_resources"] = {"error": repr(e)}
json.dump(out, open(f"{OUT}/{tag}_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
