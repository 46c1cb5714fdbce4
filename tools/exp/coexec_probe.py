"""Victim kernels of one instruction class each (tools/exp/coexec_probe.hip) on a side stream against the 1x1-only coupling_head looping on the main stream.
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libcoexec.so tools/exp/coexec_probe.hip; the resize cases need the packed product build (BFSR_HIP_LIB=$PWD/tools/exp/libpk.so).
GPU box: python tools/exp/coexec_probe.py"""
import ctypes as C, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from bfsr_amd.ops import HipOps, MODE_BILINEAR_AC, MODE_NEAREST, MODE_BILINEAR
ops = HipOps("cuda:0")
lib = C.CDLL(os.path.join(HERE, "libcoexec.so"))
g = np.random.Generator(np.random.PCG64(3))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
B = 32
hp1 = ops.pack_coupling_head(None, r(64, 64, 1, 1, scale=0.1), r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1)))
raw, h2b = torch.randn(B, 64, 96, 96, device="cuda"), ops.h2_empty(B, 64, 96, 96)
aggr = lambda: ops.coupling_head(None, hp1, raw, h2b, pre_fmt=0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
N = 1 << 24
xs = torch.randn(N, device="cuda")
o_u, o_f = torch.empty(2 * N, dtype=torch.int32, device="cuda"), torch.empty(N, device="cuda")
bottom, cat = torch.randn(B, 256, 48, 48, device="cuda"), ops.empty(B, 512, 96, 96)
b96, c192 = torch.randn(B, 64, 96, 96, device="cuda"), ops.empty(B, 64, 192, 192)
c95 = ops.empty(B, 256, 95, 95)
vict = {
    "resize bilinear-AC 48->95 (scalar kernel)": lambda: ops.resize(bottom, c95, MODE_BILINEAR_AC, 47.0 / 94.0, 47.0 / 94.0),
    "integer division by a run-time divisor (v_rcp_iflag path)": lambda: (lib.run_udiv(C.c_void_p(o_u.data_ptr()), N, 97, st()), o_u)[1],
    "float index arithmetic (cvt / floor / select)": lambda: (lib.run_index(C.c_void_p(o_f.data_ptr()), N, C.c_float(47.0 / 95.0), st()), o_f)[1],
    "four dependent gathers + blend": lambda: (lib.run_gather(C.c_void_p(xs.data_ptr()), C.c_void_p(o_f.data_ptr()), N, N - 1, st()), o_f)[1],
    "IEEE float division": lambda: (lib.run_fdiv(C.c_void_p(xs.data_ptr()), C.c_void_p(o_f.data_ptr()), N, st()), o_f)[1],
    "resize bilinear-AC 48->96 (four-wide kernel)": lambda: ops.resize(bottom, cat[:, 256:], MODE_BILINEAR_AC, 47.0 / 95.0, 47.0 / 95.0, window=(0, 0, 96, 96)),
    "resize bilinear 96->192 (four-wide kernel)": lambda: ops.resize(b96, c192, MODE_BILINEAR, 0.5, 0.5),
    "resize nearest 96->192 (four-wide kernel)": lambda: ops.resize(b96, c192, MODE_NEAREST, 0.5, 0.5),
}
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
aggr(); torch.cuda.synchronize()
for vn, vfn in vict.items():
    ref = vfn().clone(); torch.cuda.synchronize()
    assert torch.equal(ref, vfn()), vn + " alone is not deterministic"
    bad = nel = 0
    for rep in range(5):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(6):
                out = vfn()
            ev = side.record_event()
        while not ev.query():
            for _ in range(8):
                aggr()
        torch.cuda.synchronize()
        n = int((out != ref).sum())
        bad += n > 0; nel = max(nel, n)
    print("%-58s wrong in %d of 5 overlapped runs (up to %d elements)" % (vn, bad, nel), flush=True)
