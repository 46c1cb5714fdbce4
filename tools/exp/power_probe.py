"""Socket power and shader clock (rocm-smi, read-only queries) while the GPU runs: nothing, an HBM-bound kernel, the matrix-pipe-bound dense-block chain
back to back, and the same chain with idle gaps between launches -- direct evidence for DESIGN.md section 5 (round 5): the dense-block kernels run
at the package power limit with a reduced shader clock.
GPU box: python tools/exp/power_probe.py [seconds per phase = 6]"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "-P", "-g", "-M", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        pw = clk = cap = None
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl and "max" not in kl and pw is None:
                try:
                    pw = float(v)
                except ValueError:
                    pass
            if "max" in kl and "power" in kl:
                try:
                    cap = float(v)
                except ValueError:
                    pass
            if "sclk" in kl and "clock" in kl and "level" not in kl:
                try:
                    clk = float(str(v).strip("()").lower().replace("mhz", ""))
                except ValueError:
                    pass
        if clk is None:                                   # rocm-smi -g reports the DPM level; amd-smi has the current per-XCD clocks
            try:
                o2 = subprocess.run(["amd-smi", "metric", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
                vals = []

                def walk(x, key=""):
                    if isinstance(x, dict):
                        if "clk" in x and isinstance(x["clk"], dict) and "gfx" in key.lower():
                            v = x["clk"].get("value")
                            if isinstance(v, (int, float)):
                                vals.append(float(v))
                        for k, v in x.items():
                            walk(v, k)
                    elif isinstance(x, list):
                        for v in x:
                            walk(v, key)
                walk(json.loads(o2))
                if vals:
                    clk = sum(vals) / len(vals)
                card["amd-smi gfx clk (mean of XCDs)"] = clk
                if not vals:
                    card["amd-smi raw"] = o2[:700]
            except Exception as e:      # noqa: BLE001
                card["amd-smi error"] = repr(e)
        return pw, clk, cap, card
    except Exception as e:          # noqa: BLE001
        return None, None, None, {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            pw, clk, cap, _ = smi()
            self.rows.append((pw, clk, cap))
            time.sleep(0.15)


def phase(name, body):
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECS:
        body()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.stop = True
    s.join()
    rows = s.rows[len(s.rows) // 3:]                 # drop the ramp
    pws = sorted(r[0] for r in rows if r[0] is not None)
    cks = sorted(r[1] for r in rows if r[1] is not None)
    cap = next((r[2] for r in rows if r[2] is not None), None)
    med = lambda a: a[len(a) // 2] if a else float("nan")
    print("%-58s %5d iterations, %7.2f ms each | power W min/med/max %6.0f %6.0f %6.0f (limit %s) | sclk MHz min/med/max %5.0f %5.0f %5.0f  [%d samples]" % (
        name, n, dt / max(n, 1) * 1e3, pws[0] if pws else float("nan"), med(pws), pws[-1] if pws else float("nan"), cap,
        cks[0] if cks else float("nan"), med(cks), cks[-1] if cks else float("nan"), len(rows)), flush=True)


# ---- workloads --------------------------------------------------------------------------------------------------------------
B, H, NB = 16, 256, 6
ring = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5 if i == 0 else torch.zeros(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for i in range(2)]
shapes = ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
pws = [ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (ci * 9) ** 0.5), 1, lazy=True) for ci, co in shapes]        # unit gain: activations stay O(1) over the 6 blocks
epis = [ops.pack_epilogue(co, bias=torch.zeros(co)) for ci, co in shapes]
specs, cur = [], 0
for r in range(NB):
    D, Dn = ring[cur], ring[cur ^ 1]
    for i, (ci, co) in enumerate(shapes[:4]):
        specs.append(dict(x=D[:, :ci // 8], pw=pws[i], out=D[:, ci // 8: ci // 8 + 4], epi=epis[i], act=2, slope=0.2))
    specs.append(dict(x=D, pw=pws[4], out=Dn[:, :8], epi=epis[4], res1=D[:, :8], alpha1=0.2))
    cur ^= 1
chain = ops.conv_chain(specs)
x0 = torch.randn(B, 64, H, H, device="cuda") * 0.5


def chain_once():
    ops.h2_pack(x0, ring[0][:, :8])                      # fresh O(1) input every launch (0.1 ms): the data never drifts to zeros or to overflow
    chain.run()
big_a, big_b = torch.randn(16, 64, 1024, 1024, device="cuda"), torch.empty(16, 64, 1024, 1024, device="cuda")     # 4.3 GB each


def chain_back_to_back():
    for _ in range(4):
        chain_once()
    torch.cuda.synchronize()


def chain_spaced():
    chain_once()
    torch.cuda.synchronize()
    time.sleep(0.012)                                    # ~ the launch's own duration of idle time


def hbm_copy():
    for _ in range(4):
        ops.axpb_clamp(big_a, big_b, 1.0, 0.0)
    torch.cuda.synchronize()


print(json.dumps(smi()[3])[:1500], flush=True)
phase("idle (host sleeps)", lambda: time.sleep(0.05))
phase("HBM-bound: axpb_clamp over 4.3 GB (read + write)", hbm_copy)
phase("dense-block chain %dx%d^2, %d blocks per launch, back to back" % (B, H, NB), chain_back_to_back)
phase("the same chain, one launch then ~12 ms idle", chain_spaced)
phase("dense-block chain again, back to back", chain_back_to_back)
ops.check_range()
