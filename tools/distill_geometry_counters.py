#!/usr/bin/env python3
"""gpurun_out/<prefix>_geo_{trace,fetch,write,sq}/ (rocprofv3 runs of tools/prof_geometries.py) -> profiles/<tag>_geometry_counters.json.

Passes (each its own rocprofv3 run; --pmc never combined with tracing):
  trace  rocprofv3 --kernel-trace --stats --output-format csv
  fetch  rocprofv3 --pmc FETCH_SIZE          (KB; doubled for gfx950, MI355X_MICROARCH.md "HBM")
  write  rocprofv3 --pmc WRITE_SIZE          (KB)
  sq     rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
Dispatches of ce_estimate_kernel are attributed to geometries by order (gpurun_out/prof_geometries_order.json)."""
import csv, glob, json, os, statistics as st, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round3"
PFX = sys.argv[2] if len(sys.argv) > 2 else "r3"
OUT = "gpurun_out/distilled"   # gpurun merges only gpurun_out/ back; copy the files into profiles/ afterwards
os.makedirs(OUT, exist_ok=True)
order = json.load(open("gpurun_out/prof_geometries_order.json"))
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)


def dispatches(d, counter=None):
    rows = list(csv.DictReader(open(newest(f"gpurun_out/{d}/*/*_counter_collection.csv"))))
    rows = [r for r in rows if ("ce_estimate" in r["Kernel_Name"] or "ce_narrow" in r["Kernel_Name"]) and (counter is None or r["Counter_Name"] == counter)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def per_geometry(rows, field="Counter_Value"):
    out, i = [], 0
    for o in order:
        chunk = rows[i:i + o["launches"]]
        i += o["launches"]
        keep = chunk[-(o.get("keep_last", o["launches"]) - 1):]                    # first launch(es) = warm-up
        out.append(st.median(float(r[field]) for r in keep) if len(chunk) > 1 else float("nan"))
    return out


res = [dict(o) for o in order]
tr = list(csv.DictReader(open(newest(f"gpurun_out/{PFX}_geo_trace/*/*_kernel_trace.csv"))))
tr = [r for r in tr if "ce_estimate" in r["Kernel_Name"] or "ce_narrow" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Dispatch_Id"]))
i = 0
for r_, o in zip(res, order):
    chunk = tr[i:i + o["launches"]]
    i += o["launches"]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in chunk[-(o.get("keep_last", o["launches"]) - 1):]]
    r_["kernel"] = chunk[0]["Kernel_Name"].split("(")[1][-40:] if False else chunk[0]["Kernel_Name"][:70]
    r_["kernel_us_median"] = st.median(durs) / 1e3
    r_["alg_GBps"] = o["alg_bytes_per_launch"] / (st.median(durs) * 1e-9) / 1e9
    r_["alg_frac_of_8TBps"] = r_["alg_GBps"] / 8000.0
for name, d, scale in (("FETCH_SIZE", f"{PFX}_geo_fetch", 1024 * 2), ("WRITE_SIZE", f"{PFX}_geo_write", 1024)):
    for r_, v in zip(res, per_geometry(dispatches(d, name))):
        r_[("hbm_read_bytes" if name == "FETCH_SIZE" else "hbm_write_bytes")] = v * scale
for r_ in res:
    r_["hbm_traffic_bytes"] = r_["hbm_read_bytes"] + r_["hbm_write_bytes"]
    r_["traffic_over_algorithmic"] = r_["hbm_traffic_bytes"] / r_["alg_bytes_per_launch"]
    r_["traffic_GBps"] = r_["hbm_traffic_bytes"] / (r_["kernel_us_median"] * 1e-6) / 1e9
sqn = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
for c in sqn:
    try:
        for r_, v in zip(res, per_geometry(dispatches(f"{PFX}_geo_sq", c))):
            r_[c] = v
    except Exception as e:
        print("missing", c, e)
for r_ in res:
    if "SQ_WAVE_CYCLES" in r_ and r_.get("SQ_WAVE_CYCLES"):
        wc = r_["SQ_WAVE_CYCLES"]
        r_["wave_parked_frac"] = r_["SQ_WAIT_ANY"] / wc            # s_waitcnt / barrier
        r_["wave_issue_stall_frac"] = r_["SQ_WAIT_INST_ANY"] / wc
        r_["wave_issuing_frac"] = r_["SQ_ACTIVE_INST_ANY"] / wc
        # SQ_*_CYCLES count quad-cycles (MI355X_MICROARCH.md): average lifetime of a wave in shader cycles
        r_["wave_lifetime_cycles"] = 4.0 * wc / r_["SQ_WAVES"]
        if r_.get("GRBM_GUI_ACTIVE"):
            # waves resident per CU on average: wave-cycles / (kernel cycles x 256 CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            r_["resident_waves_per_cu"] = 4.0 * wc / (r_["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
json.dump(dict(passes=__doc__.split("Passes")[1].strip().splitlines()[:5], geometries=res), open(f"{OUT}/{tag}_geometry_counters.json", "w"), indent=1)
for r_ in res:
    print(f"{r_['name'][:46]:46s} {r_['kernel_us_median']:8.1f} us  alg {r_['alg_GBps']:6.0f} GB/s ({r_['alg_frac_of_8TBps']:.3f})  traffic x{r_['traffic_over_algorithmic']:.2f}"
          f"  parked {r_.get('wave_parked_frac', float('nan')):.2f} stall {r_.get('wave_issue_stall_frac', float('nan')):.2f} issuing {r_.get('wave_issuing_frac', float('nan')):.2f}"
          f"  resident waves/CU {r_.get('resident_waves_per_cu', float('nan')):.1f}")
