"""Summarise a rocprofv3 --pmc counter_collection.csv: one line per distinct (kernel, grid) with counters
normalised per wave where it helps.  Usage: python tools/pmc_report.py <dir>"""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = (r["Dispatch_Id"], r["Kernel_Name"][:70], r["Grid_Size"])
        agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = collections.OrderedDict()
    for k, v in agg.items():
        last[(k[1], k[2])] = v          # keep the last dispatch of each (kernel, grid): warmed up
    for (name, grid), v in last.items():
        if "conv" not in name and "flow_pointwise" not in name:
            continue
        print(name, "grid", grid)
        print("   ", {a: int(b) for a, b in sorted(v.items())})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"]:
            # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD-quad (4 per CU), SQ_BUSY_CYCLES per SE-level SQ: normalise by GRBM
            d = {"mfma_busy/(4*busy_cu_cycles)": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v.get("SQ_BUSY_CU_CYCLES", float("nan")))
                 if v.get("SQ_BUSY_CU_CYCLES") else None,
                 "lds_idx_active/busy_cu_cycles": v.get("SQ_LDS_IDX_ACTIVE", 0) / v["SQ_BUSY_CU_CYCLES"] if v.get("SQ_BUSY_CU_CYCLES") else None,
                 "lds_conflict/lds_active": v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else None}
            print("    derived:", d)
