import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bfsr_amd import _lib
if os.environ.get("HEADLIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["HEADLIB"])
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
C, hw, B = 12, 320, 8
cn, cc2 = C // 2, 2 * (C - C // 2)
hpk = ops.pack_coupling_head(r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1), r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1)))
tpk = ops.pack_coupling_tail(r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2)))
z = torch.randn(B, C, hw, hw, device="cuda"); pre = torch.randn(B, 64, hw, hw, device="cuda"); hid = torch.empty(B, 64, hw, hw, device="cuda")
hf = torch.randn(B, 2 * C, hw, hw, device="cuda") * 0.5
for fmt in (0, 1, 0, 1):
    print(os.environ.get("HEADLIB", "product"), "hid_fmt", fmt, "head %.1f us" % np.median([timed(lambda: ops.coupling_head(z, hpk, pre, hid, hid_fmt=fmt)) for _ in range(5)]),
          "tail %.1f us" % np.median([timed(lambda: ops.coupling_tail(hid, tpk, z, z, 1, h_ft=hf, hid_fmt=fmt)) for _ in range(5)]))
