#!/usr/bin/env python3
"""gpurun_out/{dn_prof,dn_pmc1,dn_pmc2,dn_pmc3}/ + dn_bench.log -> profiles/<tag>_denoise_summary.json (Conv2d denoiser extension).

On the GPU box (program itself after `--`, counters in their own passes):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dn_prof -- python3 bench.py --workload pusch273_4rx_denoise --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dn_bench.log
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/dn_pmc1 -- python3 tools/denoise_perf.py 512
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/dn_pmc2 -- python3 tools/denoise_perf.py 512
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/dn_pmc3 -- python3 tools/denoise_perf.py 512"""
import csv, glob, collections, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
out = {"workload": "tools/denoise_perf.py 512 (2048 planes of 3276 x 14) for the PMC passes; bench.py --workload pusch273_4rx_denoise for the kernel trace",
       "pmc_note": "separate rocprofv3 --pmc passes (4 counters each), medians over the kernel's launches; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count "
                   "quad-cycles summed over waves; GRBM_GUI_ACTIVE sums the 8 XCDs"}
for d in ("dn_pmc1", "dn_pmc2", "dn_pmc3"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"gpurun_out/{d}/*/*counter_collection.csv"))):
        if "denoise" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sorted(v)[len(v) // 2]
simd_cycles = out["GRBM_GUI_ACTIVE"] / 8 * 256 * 4
out["derived"] = {"mfma_busy_fraction_of_simd_cycles": out["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                  "lds_active_fraction_of_cu_cycles": out["SQ_LDS_IDX_ACTIVE"] / (out["GRBM_GUI_ACTIVE"] / 8 * 256),
                  "lds_bank_conflict_fraction_of_lds_active": out["SQ_LDS_BANK_CONFLICT"] / out["SQ_LDS_IDX_ACTIVE"],
                  "valu_per_mfma": out["SQ_INSTS_VALU"] / out["SQ_INSTS_MFMA"],
                  "wave_wait_fraction": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]}
ks = list(csv.DictReader(open(newest("gpurun_out/dn_prof/*/*_kernel_stats.csv"))))
out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in ks[:4]]
out["bench_line"] = json.loads([l for l in open("gpurun_out/dn_bench.log") if l.startswith("{")][-1])
json.dump(out, open(f"profiles/{tag}_denoise_summary.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1), out["kernel_stats"][0])
