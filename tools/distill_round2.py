#!/usr/bin/env python3
"""gpurun_out/r2_bench_{trace,fetch,write,sq}/ + r2_bench_same_lease.log + r2_rwmix.log (one gpurun lease,
tools/gpu_profile_round.sh) -> profiles/round2_{summary.json,kernel_stats.csv,bench_line.json}."""
import csv, glob, json, os, re, statistics as st, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round2"
OUT = "gpurun_out/distilled"   # gpurun merges only gpurun_out/ back; copy the files into profiles/ afterwards
os.makedirs(OUT, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
ks = list(csv.DictReader(open(newest("gpurun_out/r2_bench_trace/*/*_kernel_stats.csv"))))

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


for name, d in (("FETCH_SIZE", "r2_bench_fetch"), ("WRITE_SIZE", "r2_bench_write")):
    rows, vals = counter(d, name)
    out[name + "_KB_per_launch_median"] = st.median(vals); out[name + "_launches"] = len(vals)
    out["lds_block_size"] = rows[0]["LDS_Block_Size"]; out["grid_threads"] = rows[0]["Grid_Size"]; out["workgroup"] = rows[0]["Workgroup_Size"]
    out["vgpr_count"] = rows[0].get("VGPR_Count"); out["sgpr_count"] = rows[0].get("SGPR_Count"); out["scratch_size"] = rows[0].get("Scratch_Size")
fetch_b = out["FETCH_SIZE_KB_per_launch_median"] * 1024 * 2   # gfx950: FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM)
write_b = out["WRITE_SIZE_KB_per_launch_median"] * 1024
alg = 13096452096
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
        sq[c] = st.median(counter("r2_bench_sq", c)[1])
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
line = [l for l in open("gpurun_out/r2_bench_same_lease.log") if l.startswith("{")][-1]
open(f"{OUT}/{tag}_bench_line.json", "w").write(line)
bl = json.loads(line)
out["same_lease_bench"] = {"ms_per_step": bl["ms_per_step"], "kernel_ms": bl["roofline"]["kernel_ms"], "roofline_frac": bl["roofline"]["frac"],
                           "secondary": [(s["workload"], s.get("rx_layout", ""), round(s["ms_per_step"], 4), round(s["roofline"]["frac"], 4)) for s in bl.get("secondary", [])],
                           "cpu_baseline": {k: bl["cpu_baseline"][k] for k in bl.get("cpu_baseline", {}) if k != "sample"}}
rw = {}
for l in open("gpurun_out/r2_rwmix.log"):
    m = re.match(r"\s*(.*?)\s*:\s*([\d.]+) ms\s+([\d.]+) GB/s(.*)", l)
    if m:
        rw[m.group(1).strip()] = {"ms": float(m.group(2)), "GBps": float(m.group(3)), "of": m.group(4).strip()}
out["same_lease_rwmix"] = rw
json.dump(out, open(f"{OUT}/{tag}_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
