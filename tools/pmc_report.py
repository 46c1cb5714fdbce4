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
        if "conv_mfma" not in name and "flow_pointwise" not in name:
            continue
        print(name, "grid", grid)
        print("   ", {a: int(b) for a, b in sorted(v.items())})
