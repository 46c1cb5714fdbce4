"""HBM traffic per launch SHAPE from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE are too wide for one pass).
Each pass runs the same command with BFSR_KEYLOG set, so the n-th dispatch of a library kernel in the counter CSV is the
n-th logged launch key (every ops call is exactly one kernel).  Output: JSON {"kernels": {key: {...}}} for bench.py.
Usage: python tools/pmc_traffic.py <fetch_dir> <fetch_keylog.json> <write_dir> <write_keylog.json> > profiles/rNN_pmc_traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2; WRITE_SIZE
is checked against the known output size of the launch (it matches to <1 %)."""
import collections
import csv
import glob
import json
import sys

LIB = ("conv_", "flow_pointwise", "squeeze2d", "unsqueeze2d", "split2d", "standardize", "resize_kernel", "maxpool2",
       "axpb_clamp", "linf_", "patch_", "grid_sample")


def per_key(d, keylog, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(t in r["Kernel_Name"] for t in LIB):
            disp[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    keys = json.load(open(keylog))
    vals = [disp[k] for k in sorted(disp)]
    if len(vals) != len(keys):
        raise SystemExit("dispatches %d != logged launches %d" % (len(vals), len(keys)))
    agg = collections.OrderedDict()
    for k, (name, v) in zip(keys, vals):
        a = agg.setdefault(json.dumps(k), {"kernel": name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], "n": 0, "sum": 0.0})
        a["n"] += 1
        a["sum"] += v
    return agg


fetch = per_key(sys.argv[1], sys.argv[2], "FETCH_SIZE")
write = per_key(sys.argv[3], sys.argv[4], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline --no-fp32-line (one pass per counter, BFSR_KEYLOG to map dispatches to launch shapes)",
       "units": "rocprofv3 reports KB; bytes = KB * 1024; fetch corrected x2 (gfx950)", "kernels": collections.OrderedDict()}
rows = []
for k, fa in fetch.items():
    wa = write.get(k)
    if not wa or not k.startswith('["conv'):
        continue
    fb, wb = fa["sum"] / fa["n"] * 1024.0, wa["sum"] / wa["n"] * 1024.0
    rows.append((fa["n"] * (2 * fb + wb), k, {"kernel": fa["kernel"], "launches": fa["n"], "fetch_bytes_raw": fb, "write_bytes": wb,
                                             "fetch_bytes_corrected_x2": 2 * fb, "hbm_bytes_per_launch": 2 * fb + wb}))
for _, k, v in sorted(rows, reverse=True)[:12]:
    out["kernels"][k] = v
print(json.dumps(out, indent=1))
