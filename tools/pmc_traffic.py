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

LIB = ("conv_", "conv3x3_x3s", "conv3x3_h2s", "conv3x3_h2x", "conv2d_direct", "x3_pack", "x3_unpack", "h2_pack", "h2_unpack", "coupling_", "flow_pointwise", "squeeze2d", "unsqueeze2d", "split2d",
       "standardize", "resize_kernel", "resize4_kernel", "resize32_kernel", "resize_h2_kernel", "maxpool2", "axpb_clamp", "linf_", "patch_", "grid_sample", "conv1x1", "gaussian_logp", "logscale_sum",
       "resample_taps", "sqdiff_sum", "ssim_sum", "to_uint8", "channel_absmax")     # (a range_check call = channel_absmax + channel_range_test: the first stands for the key)
# FETCH_SIZE on gfx950 counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM; documented there for 16-B-per-lane streaming reads).
# Calibration for THIS library's access patterns, from the same PMC run (profiles/r02_pmc_traffic.json "calibration"): the plain 1x1
# conv 64->64 @ 8x320x320 reads its 209.7 MB input exactly once with 4-byte-per-lane row loads and reports 105.3 MB raw = 0.502, so the
# x2 reading holds for the 4-byte row loads of the conv kernels as well as for the 16-byte LDS-DMA of conv_x3s: x2 for every family.
WIDE_READERS = ("conv", "coupling", "linf", "flow")     # calibrated in r02: the 1x1 conv 64->64 reads its input once, raw FETCH_SIZE = 0.50 of it -> x2 everywhere


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
       "units": "rocprofv3 reports KB; bytes = KB * 1024; hbm_bytes_per_launch = fetch x fetch_correction (x2 on gfx950, calibrated: see the comment in tools/pmc_traffic.py) + write",
       "kernels": collections.OrderedDict()}
rows = []
for k, fa in fetch.items():
    wa = write.get(k)
    if not wa or not (k.startswith('["conv') or k.startswith('["coupling') or k.startswith('["linf_mlp') or k.startswith('["flow')):
        continue
    fb, wb = fa["sum"] / fa["n"] * 1024.0, wa["sum"] / wa["n"] * 1024.0
    corr = 2.0 if any(k.startswith('["%s' % w) for w in WIDE_READERS) else 1.0
    rows.append((fa["n"] * (corr * fb + wb), k, {"kernel": fa["kernel"], "launches": fa["n"], "fetch_bytes_raw": fb, "write_bytes": wb,
                                                "fetch_correction": corr, "hbm_bytes_per_launch": corr * fb + wb,
                                                "unit": "HBM-side bytes per launch: rocprofv3 FETCH_SIZE x %g + WRITE_SIZE" % corr}))
for _, k, v in sorted(rows, reverse=True)[:16]:
    out["kernels"][k] = v
print(json.dumps(out, indent=1))
