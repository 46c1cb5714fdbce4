"""Aggregate rocprofv3 counter_collection.csv files: per (kernel, grid) the LAST dispatch's counters, merged over passes."""
import collections, csv, glob, sys
agg = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:60], r["Grid_Size"], r.get("Workgroup_Size", ""))
            per.setdefault(k, collections.OrderedDict()).setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, disp in per.items():
            last = list(disp.values())[-1]
            agg.setdefault(k, {}).update(last)
            agg[k]["_n"] = len(disp)
for k, v in agg.items():
    if "conv" not in k[0] and "flow" not in k[0] and "linf" not in k[0] and "rdb" not in k[0]:
        continue
    print(k)
    wc = v.get("SQ_WAVE_CYCLES", 0)
    for n in sorted(v):
        if n.startswith("_"):
            continue
        s = "   %-28s %16.0f" % (n, v[n])
        if wc and n.startswith("SQ_") and ("WAIT" in n or "ACTIVE" in n):
            s += "   %5.1f%% of wave cycles" % (100.0 * v[n] / wc)
        print(s)
    if v.get("SQ_BUSY_CU_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        print("   mfma busy = %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_BUSY_CU_CYCLES"])))
    if v.get("SQ_WAVES") and v.get("SQ_INSTS_MFMA"):
        print("   per wave: mfma %.0f valu %.0f lds %.0f salu %.0f vmem_rd %.0f" % tuple(v.get(n, 0) / v["SQ_WAVES"] for n in
              ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD")))
