#!/usr/bin/env python3
"""gpurun_out/<p>_hc_{bench,rwmix}_<i>/ (tools/headline_counters.sh: one rocprofv3 --pmc pass per counter set, headline bench
and the access-pattern micro-benchmark on ONE lease) -> gpurun_out/distilled/<tag>_headline_counters.json.

rwmix rows = its first configuration only (3 workgroups/CU, XCD map on, stores depend on the loads: the estimator's access
pattern with no estimation work), i.e. the first 7 dispatches of rwmix<true, true>."""
import csv, glob, json, os, statistics as st, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round3"
PFX = sys.argv[2] if len(sys.argv) > 2 else "r3"
OUT = "gpurun_out/distilled"
os.makedirs(OUT, exist_ok=True)


def collect(kind, kern, first=None):
    res = {}
    for d in sorted(glob.glob(f"gpurun_out/{PFX}_hc_{kind}_*/")):
        for f in glob.glob(d + "*/*_counter_collection.csv"):
            rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            keep = set(ids[1:first] if first else ids[1:])          # first dispatch = warm-up
            by = {}
            for r in rows:
                if int(r["Dispatch_Id"]) in keep:
                    by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in by.items():
                res[k] = st.median(v)
    return res


def derive(c):
    d = {}
    g = c.get
    if g("SQC_ICACHE_REQ"):
        d["icache_miss_rate"] = g("SQC_ICACHE_MISSES", 0) / g("SQC_ICACHE_REQ")
        d["icache_miss_rate_incl_duplicates"] = (g("SQC_ICACHE_MISSES", 0) + g("SQC_ICACHE_MISSES_DUPLICATE", 0)) / g("SQC_ICACHE_REQ")
    if g("SQC_DCACHE_REQ"):
        d["scalar_dcache_miss_rate"] = g("SQC_DCACHE_MISSES", 0) / g("SQC_DCACHE_REQ")
    if g("TCP_TCC_READ_REQ_sum"):
        d["avg_l1_to_l2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
    if g("TCC_EA0_RDREQ_sum"):
        d["avg_l2_to_fabric_read_latency_cycles"] = g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum")
    if g("TCC_EA0_WRREQ_sum"):
        d["fabric_write_stall_cycles_per_write_request"] = g("TCC_EA0_WRREQ_STALL_sum") / g("TCC_EA0_WRREQ_sum")
    if g("TCC_REQ_sum"):
        d["l2_hit_rate"] = g("TCC_HIT_sum") / g("TCC_REQ_sum")
    if g("SQ_WAVES"):
        w = g("SQ_WAVES")
        d["instructions_per_wave"] = {k[9:].lower(): g(k) / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if g(k) is not None}
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        d["wave_states"] = {"parked_s_waitcnt_or_barrier": g("SQ_WAIT_ANY", 0) / wc, "issue_stalled": g("SQ_WAIT_INST_ANY", 0) / wc, "issuing": g("SQ_ACTIVE_INST_ANY", 0) / wc}
    return d


est = collect("bench", "ce_estimate")
rw = collect("rwmix", "rwmix<true, true>", first=7)
out = {"what": __doc__.strip().splitlines()[0], "passes": "tools/headline_counters.sh (one rocprofv3 --pmc pass per set; never combined with tracing)",
       "estimator_kernel": {"counters": est, "derived": derive(est)},
       "rwmix_3wg_xcdmap_dependent_write": {"counters": rw, "derived": derive(rw)}}
json.dump(out, open(f"{OUT}/{tag}_headline_counters.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items() if isinstance(v, dict)}, indent=1))
