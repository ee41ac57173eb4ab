#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs a gpurun call left under gpurun_out/ into the small, tracked files under profiles/.
Usage: python tools/distill_profiles.py [round_tag]   (expects gpurun_out/{prof_r1,pmc_fetch,pmc_write}/ and bench_full.log)"""
import csv, glob, json, os, statistics as st, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
ks = list(csv.DictReader(open(newest("gpurun_out/prof_r1/*/*_kernel_stats.csv"))))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(ks[0].keys())
    for r in ks[:12]:
        r = dict(r); r["Name"] = r["Name"][:160]; w.writerow(r.values())
ce = [r for r in ks if "ce_estimate" in r["Name"]][0]
out = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline",
       "kernel": ce["Name"][:80], "calls": int(ce["Calls"]), "avg_ns": float(ce["AverageNs"]), "min_ns": int(ce["MinNs"]), "max_ns": int(ce["MaxNs"])}
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    rows = [r for r in csv.DictReader(open(newest(f"gpurun_out/{d}/*/*_counter_collection.csv"))) if "ce_estimate" in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals = [float(r["Counter_Value"]) for r in rows]
    out[name + "_KB_per_launch_median"] = st.median(vals); out[name + "_launches"] = len(vals)
    out["lds_block_size"] = rows[0]["LDS_Block_Size"]; out["grid_threads"] = rows[0]["Grid_Size"]; out["workgroup"] = rows[0]["Workgroup_Size"]
fetch_b = out["FETCH_SIZE_KB_per_launch_median"] * 1024 * 2   # gfx950: FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM)
write_b = out["WRITE_SIZE_KB_per_launch_median"] * 1024
out.update(hbm_read_bytes_per_launch_corrected=fetch_b, hbm_write_bytes_per_launch=write_b, hbm_traffic_bytes_per_launch=fetch_b + write_b,
           algorithmic_bytes_per_launch=13096452096,
           pmc_commands=["rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                         "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"],
           workload="pusch273_4rx_filter (8192 slots x 4 ports, 273 PRB)")
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
line = [l for l in open("gpurun_out/bench_full.log") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench_line.json", "w").write(line)
print(json.dumps(out, indent=1)); print(line[:300])
