"""Phase timeline of coupling_step_kernel (workgroup 0, every wave, first tiles) from the s_memtime stamps of the trace build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bfsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STEPLIB", "libstep.so"))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
ops.lib.bfsr_debug_step_trace.restype = ctypes.c_int
ops.lib.bfsr_debug_step_trace.argtypes = [ctypes.c_void_p]
C = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 320
B = 8
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
cn, cc2 = C // 2, 2 * (C - C // 2)
spk = ops.pack_coupling_step(r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1), r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1),
                             torch.exp(r(64, scale=0.1)), r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2)))
Wm = ops.vec(torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32)))
bias, es = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
z = torch.randn(B, C, hw, hw, device="cuda"); z2 = torch.empty_like(z)
pre = torch.randn(B, 64, hw, hw, device="cuda") * 0.5
hf = torch.randn(B, 2 * C, hw, hw, device="cuda") * 0.5
run = lambda: ops.coupling_step(z, z2, spk, pre, 1, h_ft=hf, w=Wm, an_bias=bias, an_escale=es)
for _ in range(3):
    run()
tr = torch.zeros(8 * 8 * 16, dtype=torch.int64, device="cuda")
assert ops.lib.bfsr_debug_step_trace(tr.data_ptr()) == 0
run(); torch.cuda.synchronize()
ops.lib.bfsr_debug_step_trace(None)
t = tr.cpu().numpy().reshape(8, 8, 16)
names = ["stage z1", "wait b1", "S1", "E1", "S2", "wait b2", "E2", "wait b3", "S3", "E3", "wait b4", "PW-A", "wait b5", "PW-B"]
print("C=%d %dx%d: cycles (s_memtime ticks = 100 MHz? see total) per phase, tile iterations 1..4 of workgroup 0" % (C, hw, hw))
import time
def timed(n=20):
    torch.cuda.synchronize(); s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(n): run()
    e_.record(); torch.cuda.synchronize()
    return s_.elapsed_time(e_) / n * 1e3
print("kernel time %.1f us (%s)" % (timed(), os.environ.get("STEPLIB", "libstep.so")))
NT = int(os.environ.get("STEP_TILES", "2"))
for it in range(2, 2 + NT):
    if t[it, 0, 0] == 0:
        break
    print("tile %d (start +%d since previous tile start)" % (it, t[it, 0, 0] - t[it - 1, 0, 0]))
    for w in (0, 3, 4, 7):
        d = [int(t[it, w, k + 1] - t[it, w, k]) for k in range(14)]
        print("  wave %d: " % w + "  ".join("%s %d" % (n, v) for n, v in zip(names, d)) + "   | total %d" % (t[it, w, 14] - t[it, w, 0]))
