#!/usr/bin/env python3
"""gpurun_out/<prefix>_{a,b}/ (two rocprofv3 --pmc passes of tools/prof_narrow.py) -> one line of SQ figures per narrow geometry
(profiles/roundN_narrow_counters.txt).   python tools/distill_narrow_counters.py <prefix, e.g. r3n>"""
import csv, glob, statistics as st, collections, sys
pfx=sys.argv[1]
res=collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/{pfx}_*/*/*_counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    by=collections.defaultdict(list)
    for r in rows:
        if "ce_" in r["Kernel_Name"]:
            by[(int(r["Dispatch_Id"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    ids=sorted({k[0] for k in by})
    # 61 launches per geometry
    for gi in range(len(ids)//61):
        chunk=ids[gi*61:(gi+1)*61][-20:]
        for cn in {k[1] for k in by}:
            vals=[sum(by[(i,cn)]) for i in chunk if (i,cn) in by]
            res[(gi,cn)]=st.median(vals)
names=["L1 2h 12PRB","L2 2h 12PRB","case4","case8 L2","case0","L2 25PRB"]
for gi,n in enumerate(names):
    g=lambda c: res.get((gi,c),float('nan'))
    w=g("SQ_WAVES")
    print(f"{n:12s} waves {w:.0f} VALU/wave {g('SQ_INSTS_VALU')/w:.0f} SALU {g('SQ_INSTS_SALU')/w:.0f} LDS {g('SQ_INSTS_LDS')/w:.0f} VMEMrd {g('SQ_INSTS_VMEM_RD')/w:.1f} wr {g('SQ_INSTS_VMEM_WR')/w:.1f} | wave cycles/wave {4*g('SQ_WAVE_CYCLES')/w:.0f} parked {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.2f} stall {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.2f} issuing {g('SQ_ACTIVE_INST_ANY')/g('SQ_WAVE_CYCLES'):.2f} valu-active {g('SQ_ACTIVE_INST_VALU')/g('SQ_WAVE_CYCLES'):.2f} busy {g('SQ_BUSY_CYCLES'):.3g}")
