"""Thread sweep of bench.py's `cpu_baseline` leg (the torch-CPU oracle of BASELINE config 2 on one 160x160 crop) on the GPU box's
host: justifies the thread count bench.py uses.  Usage (GPU box):  python tools/cpu_thread_sweep.py > gpurun_out/cpu_thread_sweep.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bfsr_amd import synth                       # noqa: E402
from bfsr_amd.srflow import options, spec        # noqa: E402
import oracle.srflow_ref as O                    # noqa: E402  (the checker being timed: this tool is measurement infrastructure)


def main():
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    nb = opt["network_G"]["nb"]
    x = synth.lr_batch(99, 1, 160, 160)
    ncpu = os.cpu_count() or 1
    rows = []
    for nt in [t for t in (8, 16, 32, 64, 128, 256) if t <= ncpu] + ([ncpu] if ncpu not in (8, 16, 32, 64, 128, 256) else []):
        torch.set_num_threads(nt)
        O.lp_pipeline(x, sd, psd, opt, nb)                  # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.lp_pipeline(x, sd, psd, opt, nb)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[1]
        rows.append({"threads": nt, "median_s": round(med, 3), "runs_s": [round(t, 3) for t in ts], "HR_MPix_per_s": round(0.4096 / med, 5)})
        print("threads %d: %.2f s" % (nt, med), file=sys.stderr, flush=True)
        print(json.dumps({"partial": rows[-1]}), flush=True)          # one line per point: a cut-off run still leaves its data
    best = min(rows, key=lambda r: r["median_s"])
    print(json.dumps({"host_cpu_count": ncpu, "workload": "oracle lp_pipeline, SRFlow-LP 4x, 1 crop 160x160 -> 640x640, fp32", "sweep": rows,
                      "best_threads": best["threads"]}, indent=1))


if __name__ == "__main__":
    main()
