"""Python face of the HIP kernels: thin wrappers that turn torch CUDA(ROCm) tensors into the
(pointer, batch stride, dims) views of the C ABI and launch on torch's current stream.

PyTorch is plumbing here (device memory, streams); all arithmetic on the hot path happens in
libbfsr_hip.so.  `HipOps` is the only ops backend of the product; tests may substitute a CPU
test double with the same interface to exercise the host-side schedule without a GPU.
"""
import ctypes as C
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
MODE_NEAREST, MODE_BILINEAR, MODE_BILINEAR_AC = 0, 1, 2


def _view(t, name="tensor"):
    """(ptr, batch_stride, C, H, W) of an NCHW fp32 view with contiguous planes."""
    if t.dtype != torch.float32 or t.dim() != 4:
        raise ValueError("%s: need a 4-d float32 tensor" % name)
    B, Cc, H, W = t.shape
    s = t.stride()
    if not (s[3] == 1 and s[2] == W and (Cc == 1 or s[1] == H * W)):
        raise ValueError("%s: not an NCHW plane-contiguous view (shape %s stride %s)" % (name, tuple(t.shape), s))
    bs = s[0] if B > 1 else Cc * H * W
    return t.data_ptr(), bs, Cc, H, W


def _ptr(t):
    return None if t is None else t.data_ptr()


class PackedConv(object):
    """A conv weight packed for the MFMA kernel (layout private to the library).  Packings for other M-tile counts
    are produced lazily from the kept OIHW copy: the launcher picks the M tile per call from the grid size."""
    __slots__ = ("_data", "Cout", "Cin", "KS", "mtile", "fixed", "_w", "_alts", "_ops", "scale", "arith")

    def __init__(self, data, Cout, Cin, KS, mtile, fixed=False, w=None, ops=None, scale=1.0, arith=0):
        # `data` may be a thunk: the base packing is then produced on first use (weights that only ever run on another kernel's packing --
        # the RRDB convs on conv_h2x -- never materialise it)
        self._data, self.Cout, self.Cin, self.KS, self.mtile, self.fixed = data, Cout, Cin, KS, mtile, fixed
        self._w, self._alts, self._ops = w, ({} if callable(data) else {mtile: data}), ops
        # arith 1: two-term fp16 split of w*scale (scale = a power of two); the kernels multiply their accumulators by 1/scale
        self.scale, self.arith = scale, arith

    @property
    def data(self):
        if callable(self._data):
            self._data = self._data()
            self._alts[self.mtile] = self._data
        return self._data

    def variant(self, mtile, packer=None):
        if mtile == self.mtile:
            return self.data
        if mtile not in self._alts:
            self._alts[mtile] = (packer or self._ops._pack_raw)(self._w, mtile)
        return self._alts[mtile]


def default_mtile(Cout):
    t = (Cout + 31) // 32
    return 1 if t <= 1 else (3 if t == 3 else 2)


class ConvChain(object):
    """A prepared conv chain (HipOps.conv_chain): host + device copies of the descriptor table, the per-tile progress words."""

    def __init__(self, ops, table_host, table_dev, progress, keep, key):
        self.ops, self.table_host, self.table_dev, self.progress, self.keep, self.key = ops, table_host, table_dev, progress, keep, key

    def run(self, tune=0):
        ops = self.ops
        _lib.check(ops._launch(self.key, lambda: ops.lib.bfsr_conv_chain_launch(
            self.table_host.data_ptr(), self.table_dev.data_ptr(), self.progress.data_ptr(), ops.range_flag.data_ptr(), tune, ops._stream())),
            "conv_chain_launch")


class HipOps(object):
    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("bfsr_amd: no GPU visible -- the engine has no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        # optional in-situ timing: launches whose key is in `profile_keys` are bracketed by HIP events on the
        # launch stream; bench.py reads `profile` (key -> [(start, end), ...]) after synchronising.
        self.profile_keys, self.profile = None, {}
        # contraction mode the engines pick for their large 3x3 convs: "x3" = fp32-accurate split on the 16-bit matrix pipe (which
        # split: `self.split` below; the API keeps its round-1 names conv_x3 / pack_conv_x3 / x3_* for both), i.e. formerly the 3xBF16 split on the bf16
        # MFMA (default), "f32" = native fp32 MFMA everywhere (BFSR_CONV=f32)
        # BFSR_KEYLOG=<path>: record the key of every launch, in order, and dump them at exit (tools/pmc_traffic.py aligns
        # them with the dispatch order of a rocprofv3 --pmc run to attribute counters to launch shapes)
        self._keylog = None
        if os.environ.get("BFSR_KEYLOG"):
            import atexit
            import json
            self._keylog = []
            atexit.register(lambda path=os.environ["BFSR_KEYLOG"], log=self._keylog: json.dump(log, open(path, "w")))
        self.conv_mode = os.environ.get("BFSR_CONV", "x3")
        if self.conv_mode not in ("x3", "f32"):
            raise ValueError("BFSR_CONV must be 'x3' or 'f32'")
        # which split the "x3" (fp32-accurate, 16-bit matrix pipe) mode uses: "f16x2" = two-term fp16 split of both operands, three
        # products, weights pre-scaled by a power of two (22 significant bits per operand; end to end indistinguishable from fp32,
        # half the matrix instructions) or "bf16x3" = the exact three-term bf16 split, six products (no range restriction on activations)
        self.split = os.environ.get("BFSR_SPLIT", "f16x2")
        if self.split not in ("f16x2", "bf16x3"):
            raise ValueError("BFSR_SPLIT must be 'f16x2' or 'bf16x3'")
        # device word the fp16-split kernels raise when an operand leaves their range (check_range())
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        # guard.run_guarded: passes re-run under the bf16x3 split because the fp16 pair met a value outside its range (0 in normal operation)
        self.fallbacks, self._fallback, self._guard_depth, self._range_scratch = 0, None, 0, None

    def cu_count(self):
        """Compute units of the device (the persistent kernels launch one workgroup per CU)."""
        return torch.cuda.get_device_properties(self.device).multi_processor_count

    def _launch(self, key, fn):
        if self._keylog is not None:
            self._keylog.append(list(key))
        if self.profile_keys is None or (self.profile_keys != "ALL" and key not in self.profile_keys):
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream(self.device)
        s.record(st)
        r = fn()
        e.record(st)
        self.profile.setdefault(key, []).append((s, e))
        return r

    # ---- memory ---------------------------------------------------------------------------
    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def to_device(self, t):
        """Device fp32 contiguous version of t; returns t ITSELF when it already is one (the engines' caches are
        keyed on tensor identity, so the same LR tensor passed to encode and decode shares its conditioning)."""
        if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad:
            return t
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- conv -----------------------------------------------------------------------------
    def _pack_raw(self, w, mtile):
        Cout, Cin, KS, _ = w.shape
        n = self.lib.bfsr_conv_packed_size(Cout, Cin, KS, mtile)
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(self.lib.bfsr_pack_conv_weight(w.data_ptr(), Cout, Cin, KS, mtile, packed.data_ptr()), "pack_conv_weight")
        return packed.to(self.device)

    def pack_conv(self, w, mtile=None, out_perm=None):
        """w: [Cout,Cin,KS,KS] (any device).  mtile=None lets the launcher choose 32-wide cout tiles per workgroup
        at call time (2 by default, 1 when the grid would otherwise be too small to fill the chip)."""
        w = w.detach().to("cpu", torch.float32).contiguous()
        if out_perm is not None:
            w = w[torch.as_tensor(out_perm, dtype=torch.long)].contiguous()
        Cout, Cin, KS, _ = w.shape
        fixed = mtile is not None
        mtile = mtile or default_mtile(Cout)
        return PackedConv(self._pack_raw(w, mtile), Cout, Cin, KS, mtile, fixed=fixed, w=w, ops=self)

    def pack_conv_f16(self, w, mtile=None, kind="f16", lazy=False):
        """fp16 packing for conv_f16 (the reduced-precision MFMA path); rounding = RNE like the kernel's staging.  lazy: the packing of the
        register-staged kernel is only produced if that kernel is ever launched with these weights."""
        w = w.detach().to("cpu", torch.float32).contiguous()
        Cout, Cin, KS, _ = w.shape
        fixed = mtile is not None or Cout <= 32
        mtile = min(mtile or 2, 2) if Cout > 32 else 1
        if kind == "bf16x3" and self.split == "f16x2":
            scale = self.pow2_scale(w)
            base = (lambda: self._pack_raw_16(w, mtile, "f16x2", scale)) if lazy else self._pack_raw_16(w, mtile, "f16x2", scale)
            return PackedConv(base, Cout, Cin, KS, mtile, fixed=fixed, w=w, ops=self, scale=scale, arith=1)
        base = (lambda: self._pack_raw_16(w, mtile, kind)) if lazy else self._pack_raw_16(w, mtile, kind)
        return PackedConv(base, Cout, Cin, KS, mtile, fixed=fixed, w=w, ops=self)

    def _pack_raw_16(self, w, mtile, kind, scale=1.0):
        Cout, Cin, KS, _ = w.shape
        if kind == "f16x2":
            packed = torch.empty(self.lib.bfsr_conv_packed_size_taps_f16x2(Cout, Cin, KS * KS, mtile), dtype=torch.int16)
            _lib.check(self.lib.bfsr_pack_conv_weight_taps_f16x2(w.data_ptr(), Cout, Cin, KS * KS, mtile, scale, packed.data_ptr()), "pack_f16x2")
            return packed.to(self.device)
        size_fn = getattr(self.lib, "bfsr_conv_packed_size_" + kind)
        pack_fn = getattr(self.lib, "bfsr_pack_conv_weight_" + kind)
        packed = torch.empty(size_fn(Cout, Cin, KS, mtile), dtype=torch.int16)
        _lib.check(pack_fn(w.data_ptr(), Cout, Cin, KS, mtile, packed.data_ptr()), "pack_" + kind)
        return packed.to(self.device)

    def pack_conv_x3(self, w, mtile=None, lazy=False):
        """Split packing for conv_x3 (fp32-accurate contraction on the 16-bit MFMA): two-term fp16 planes of w * 2^k under
        split == "f16x2" (the default), three-term bf16 planes under "bf16x3".  lazy: see pack_conv_f16 (the conv_x3s / conv_h2x
        packings are always derived on first use)."""
        return self.pack_conv_f16(w, mtile, kind="bf16x3", lazy=lazy)

    def conv_x3(self, x, pw, out, **kw):
        """Same contract as conv(); fp32 operands split exactly into 3 bf16 terms, 6 cross products accumulated in fp32."""
        return self.conv_f16(x, pw, out, _kind="bf16x3", **kw)

    def conv_f16(self, x, pw, out, in_shift=0, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0,
                 res2=None, alpha2=1.0, tune=0, _kind="f16", y_fmt=0):
        """Same contract as conv(), contraction in fp16 on the 16x faster MFMA (inputs/weights rounded to fp16)."""
        xp, xbs, Cin, Hs, Ws = _view(x, "conv_f16.x")
        yp, ybs, Cout, H, W = _view(out, "conv_f16.out")
        if Cin != pw.Cin or Cout != pw.Cout or (Hs << in_shift) != H or (Ws << in_shift) != W or x.shape[0] != out.shape[0]:
            raise ValueError("conv_f16: shape mismatch x%s out%s" % (tuple(x.shape), tuple(out.shape)))
        a = _lib.BfsrConvArgs()
        a.x, a.x_bs, a.Cin = xp, xbs, Cin
        mtile, wdata = pw.mtile, pw.data
        if not pw.fixed and mtile == 2 and pw.KS == 3 and _kind == "bf16x3":
            # small grids: 32-cout workgroups (two per CU, more of them) beat the 64-cout tile (measured on MI355X:
            # 64->96 @ 8x80x80 59 -> 49 us, 64->48 @ 8x160x160 106 -> 97 us, RDB conv5 @ 8x160x160 ~ -8 %)
            if ((W + 31) // 32) * ((H + 7) // 8) * out.shape[0] * ((Cout + 31) // 32) < 2000:
                mtile = 1
                wdata = pw.variant(1, lambda w_, m_: self._pack_raw_16(w_, m_, "f16x2" if pw.arith == 1 else _kind, pw.scale))
        a.w = wdata.data_ptr()
        if pw.arith == 1:
            if _kind != "bf16x3":
                raise ValueError("conv_f16: a two-term fp16 split packing belongs to conv_x3")
            a.arith, a.acc_scale, a.flag = 1, 1.0 / pw.scale, self.range_flag.data_ptr()
        a.y, a.y_bs, a.Cout = yp, ybs, Cout
        a.B, a.H, a.W, a.KS, a.in_shift, a.mtile = out.shape[0], H, W, pw.KS, in_shift, mtile
        a.epi, a.act, a.slope = _ptr(epi), act, slope
        for name, t, al in (("pre_add", pre_add, None), ("res1", res1, alpha1), ("res2", res2, alpha2)):
            if t is not None:
                pp, bs, c, hh, ww = _view(t, "conv_f16." + name)
                assert (c, hh, ww) == (Cout, H, W)
                setattr(a, name, pp)
                setattr(a, name + "_bs", bs)
                if al is not None:
                    setattr(a, "alpha" + name[-1], al)
        a.tune = tune
        if y_fmt:                               # quad-major out / pre_add ([B][Cout/4][H][W][4] in the same buffers): split-conv kernels only
            if _kind != "bf16x3":
                raise ValueError("conv_f16: y_fmt=1 is a conv_x3 feature")
            a.y_fmt = 1
        key = ("conv_" + ("f16x2" if pw.arith == 1 else _kind), pw.KS, mtile, Cin, Cout, out.shape[0], H, W)
        fn = getattr(self.lib, "bfsr_conv2d_" + _kind)
        _lib.check(self._launch(key, lambda: fn(C.byref(a), self._stream())), "conv2d_" + _kind)
        return out

    def pack_conv1x1(self, w, x3=True):
        """Wide 1x1 conv weights ([Cout,Cin,1,1] or [Cout,Cin]) for conv1x1: x3 = exact bf16 triple, else fp16."""
        w = w.detach().to("cpu", torch.float32).reshape(w.shape[0], w.shape[1]).contiguous()
        Cout, Cin = w.shape
        packed = torch.empty(self.lib.bfsr_conv1x1_packed_size(Cout, Cin, int(x3)), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv1x1_weight(w.data_ptr(), Cout, Cin, int(x3), packed.data_ptr()), "pack_conv1x1")
        return PackedConv(packed.to(self.device), Cout, Cin, 1, 8, fixed=True)

    def conv1x1(self, x, pw, out, x3=True, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0, res2=None,
                alpha2=1.0):
        """1x1 conv as a GEMM over the flattened pixels, 256 output channels per workgroup (conv1x1.hip)."""
        xp, xbs, Cin, H, W = _view(x, "conv1x1.x")
        yp, ybs, Cout, H2, W2 = _view(out, "conv1x1.out")
        if (Cin, Cout, H, W) != (pw.Cin, pw.Cout, H2, W2) or x.shape[0] != out.shape[0] or pw.KS != 1:
            raise ValueError("conv1x1: shape mismatch x%s out%s" % (tuple(x.shape), tuple(out.shape)))
        a = _lib.BfsrConvArgs()
        a.x, a.x_bs, a.Cin = xp, xbs, Cin
        a.w = pw.data.data_ptr()
        a.y, a.y_bs, a.Cout = yp, ybs, Cout
        a.B, a.H, a.W, a.KS, a.mtile = out.shape[0], H, W, 1, pw.mtile
        a.epi, a.act, a.slope = _ptr(epi), act, slope
        for name, t, al in (("pre_add", pre_add, None), ("res1", res1, alpha1), ("res2", res2, alpha2)):
            if t is not None:
                pp, bs, c, hh, ww = _view(t, "conv1x1." + name)
                assert (c, hh, ww) == (Cout, H, W)
                setattr(a, name, pp)
                setattr(a, name + "_bs", bs)
                if al is not None:
                    setattr(a, "alpha" + name[-1], al)
        key = ("conv1x1_x3" if x3 else "conv1x1_f16", Cin, Cout, out.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv1x1(C.byref(a), int(x3), self._stream())), "conv1x1")
        return out

    @staticmethod
    def presum_up2_weights(w):
        """[Cout,Cin,3,3] -> [Cout,Cin,16]: the 3x3 taps folded onto the 2x2 source pixels each output parity of a
        nearest-x2-upsampled input touches (tap t = (a*2+b)*4 + i*2+j; include/bfsr_hip.h bfsr_conv2d_up2)."""
        R = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}       # (parity, source offset) -> kernel rows
        w = w.detach().to("cpu", torch.float32)
        out = torch.zeros(w.shape[0], w.shape[1], 16, dtype=torch.float32)
        for a in (0, 1):
            for b in (0, 1):
                for i in (0, 1):
                    for j in (0, 1):
                        acc = None
                        for dy in R[(a, i)]:
                            for dx in R[(b, j)]:
                                acc = w[:, :, dy, dx].clone() if acc is None else acc + w[:, :, dy, dx]
                        out[:, :, (a * 2 + b) * 4 + i * 2 + j] = acc
        return out.contiguous()

    def pack_conv_up2(self, w, mtile=2):
        """Pack a 3x3 weight for conv_up2 (conv over a nearest-x2-upsampled input at 4/9 of the MACs)."""
        w16 = self.presum_up2_weights(w)
        Cout, Cin, _ = w16.shape
        n = self.lib.bfsr_conv_packed_size_taps(Cout, Cin, 16, mtile)
        packed = torch.empty(n, dtype=torch.float32)
        _lib.check(self.lib.bfsr_pack_conv_weight_taps(w16.data_ptr(), Cout, Cin, 16, mtile, packed.data_ptr()), "pack_taps")
        return PackedConv(packed.to(self.device), Cout, Cin, 3, mtile, fixed=True)

    def conv_up2(self, x, pw, out, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, key=None):
        """out [B,Cout,2h,2w] = epilogue(conv3x3(nearest_up2(x [B,Cin,h,w]))).
        key = (x2 [B,Cin2,2h,2w], PackedConv 3x3 with the same mtile): extra input channels at output resolution,
        i.e. the conv over cat[x2, nearest_up2(x)] in one kernel."""
        xp, xbs, Cin, h, w = _view(x, "conv_up2.x")
        yp, ybs, Cout, H, W = _view(out, "conv_up2.out")
        if (Cin, Cout, 2 * h, 2 * w) != (pw.Cin, pw.Cout, H, W) or x.shape[0] != out.shape[0]:
            raise ValueError("conv_up2: shape mismatch x%s out%s" % (tuple(x.shape), tuple(out.shape)))
        a = _lib.BfsrConvArgs()
        a.x, a.x_bs, a.Cin = xp, xbs, Cin
        a.w = pw.data.data_ptr()
        a.y, a.y_bs, a.Cout = yp, ybs, Cout
        a.B, a.H, a.W, a.KS, a.mtile = out.shape[0], H, W, 3, pw.mtile
        a.epi, a.act, a.slope = _ptr(epi), act, slope
        if pre_add is not None:
            pp, bs, c, hh, ww = _view(pre_add, "conv_up2.pre_add")
            assert (c, hh, ww) == (Cout, H, W)
            a.pre_add, a.pre_add_bs = pp, bs
        cin2 = 0
        if key is not None:
            x2, pk = key
            a.x2, a.x2_bs, cin2, h2, w2 = _view(x2, "conv_up2.x2")
            if (h2, w2) != (H, W) or pk.Cin != cin2 or pk.Cout != Cout or pk.KS != 3 or pk.mtile != pw.mtile:
                raise ValueError("conv_up2: bad key input")
            a.Cin2, a.w_x2 = cin2, pk.data.data_ptr()
        key = ("conv_up2", pw.mtile, Cin, Cout, out.shape[0], H, W, cin2)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up2(C.byref(a), self._stream())), "conv2d_up2")
        return out

    def pack_conv_up2_x3(self, w):
        """conv_up2 weights (16 parity-pre-summed matrices) in the split layout of `self.split`."""
        w16 = self.presum_up2_weights(w)
        return self._pack_taps_x3(w16, 16)

    def _pack_taps_x3(self, wt, T):
        Cout, Cin, _ = wt.shape
        if self.split == "f16x2":
            scale = self.pow2_scale(wt)
            mt = 1      # 64-cout workgroups for the x2 taps kernel (round 3, BFSR_TAPS_MT=2) were measured slower (8.0 vs 6.5 ms) and are gone
            packed = torch.empty(self.lib.bfsr_conv_packed_size_taps_f16x2(Cout, Cin, T, mt), dtype=torch.int16)
            _lib.check(self.lib.bfsr_pack_conv_weight_taps_f16x2(wt.data_ptr(), Cout, Cin, T, mt, scale, packed.data_ptr()), "pack_taps_f16x2")
            return PackedConv(packed.to(self.device), Cout, Cin, 3, mt, fixed=True, scale=scale, arith=1, w=wt.contiguous(), ops=self)
        packed = torch.empty(self.lib.bfsr_conv_packed_size_taps_bf16x3(Cout, Cin, T, 1), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv_weight_taps_bf16x3(wt.data_ptr(), Cout, Cin, T, 1, packed.data_ptr()), "pack_taps_x3")
        return PackedConv(packed.to(self.device), Cout, Cin, 3, 1, fixed=True)

    @staticmethod
    def presum_up4_weights(w):
        """[Cout,Cin,3,3] -> [Cout,Cin,25]: per axis 5 entries (phase class, source offset) <- kernel taps:
        (0,-1)<-{-1}  (0,0)<-{0,+1}  (12,0)<-{-1,0,+1}  (3,0)<-{-1,0}  (3,+1)<-{+1}; index = row_entry*5 + col_entry."""
        S = [(0,), (1, 2), (0, 1, 2), (0, 1), (2,)]
        w = w.detach().to("cpu", torch.float32)
        out = torch.zeros(w.shape[0], w.shape[1], 25, dtype=torch.float32)
        for er in range(5):
            for ec in range(5):
                acc = None
                for dy in S[er]:
                    for dx in S[ec]:
                        acc = w[:, :, dy, dx].clone() if acc is None else acc + w[:, :, dy, dx]
                out[:, :, er * 5 + ec] = acc
        return out.contiguous()

    def pack_conv_up4_x3(self, w):
        w25 = self.presum_up4_weights(w)
        return self._pack_taps_x3(w25, 25)

    def conv_up4_x3(self, x, pw, out, epi=None, pre_add=None, act=ACT_NONE, slope=0.2):
        """out [B,Cout,4h,4w] = epilogue(conv3x3(nearest_up4(x)) + pre_add) on the split contraction (25 pre-summed matrices)."""
        return self.conv_up2_x3(x, pw, out, epi=epi, pre_add=pre_add, act=act, slope=slope, _factor=4)

    def conv_up2_x3(self, x, pw, out, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, tune=0, _factor=2, y_fmt=0):
        """conv_up2 on the split contraction (fp32-accurate; the split is the one `pw` was packed with); channels at output resolution
        enter through pre_add (may be `out`)."""
        xp, xbs, Cin, h, w = _view(x, "conv_up2_x3.x")
        yp, ybs, Cout, H, W = _view(out, "conv_up2_x3.out")
        if (Cin, Cout, _factor * h, _factor * w) != (pw.Cin, pw.Cout, H, W) or x.shape[0] != out.shape[0]:
            raise ValueError("conv_up%d_x3: shape mismatch x%s out%s" % (_factor, tuple(x.shape), tuple(out.shape)))
        a = _lib.BfsrConvArgs()
        a.x, a.x_bs, a.Cin = xp, xbs, Cin
        a.w = pw.data.data_ptr()
        a.y, a.y_bs, a.Cout = yp, ybs, Cout
        a.B, a.H, a.W, a.KS, a.mtile, a.tune = out.shape[0], H, W, 3, pw.mtile, tune
        a.epi, a.act, a.slope = _ptr(epi), act, slope
        if pw.arith == 1:
            a.arith, a.acc_scale, a.flag = 1, 1.0 / pw.scale, self.range_flag.data_ptr()
        fam = "_f2" if pw.arith == 1 else "_x3"
        if pre_add is not None:
            pp, bs, c, hh, ww = _view(pre_add, "conv_up2_x3.pre_add")
            assert (c, hh, ww) == (Cout, H, W)
            a.pre_add, a.pre_add_bs = pp, bs
        a.y_fmt = int(bool(y_fmt))              # quad-major out / pre_add (x2 kernel only; the x4 launcher rejects it)
        if _factor == 4:
            key = ("conv_up4" + fam, 1, Cin, Cout, out.shape[0], H, W, 0)
            _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up4_bf16x3(C.byref(a), self._stream())), "conv2d_up4_bf16x3")
            return out
        key = ("conv_up2" + fam, 1, Cin, Cout, out.shape[0], H, W, 0)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up2_bf16x3(C.byref(a), self._stream())), "conv2d_up2_bf16x3")
        return out

    # ---- split tensors: x3 (exact 3-term bf16 split, conv_x3s.hip) under split == "bf16x3", h2 (fp16 hi + lo, conv3x3_h2x_kernel in
    # conv_h2s.hip) under "f16x2"; x3_empty / x3_pack / x3_unpack / conv_x3s dispatch on the mode and the tensor type ------------
    def x3_empty(self, B, Cc, H, W):
        """[B, C/8, 3, H, W, 8] bf16: x = h + m + l exactly; channel slices `t[:, a//8:b//8]` are views."""
        if Cc % 8:
            raise ValueError("x3 tensors need a multiple of 8 channels")
        if self.split == "f16x2":                                      # the split tensors of this mode are h2 tensors (hi + lo fp16 planes)
            return self.h2_empty(B, Cc, H, W)
        return torch.empty(B, Cc // 8, 3, H, W, 8, dtype=torch.bfloat16, device=self.device)

    @staticmethod
    def _x3view(t, name="x3"):
        """(ptr, batch stride in bf16 elements, C, H, W) of an x3 view."""
        if t.dtype != torch.bfloat16 or t.dim() != 6 or t.shape[2] != 3 or t.shape[5] != 8:
            raise ValueError("%s: need a [B,C/8,3,H,W,8] bfloat16 tensor" % name)
        B, C8, _, H, W, _ = t.shape
        st = t.stride()
        if not (st[5] == 1 and st[4] == 8 and st[3] == 8 * W and st[2] == 8 * W * H and (C8 == 1 or st[1] == 24 * W * H)):
            raise ValueError("%s: not an x3 view (shape %s stride %s)" % (name, tuple(t.shape), st))
        return t.data_ptr(), (st[0] if B > 1 else C8 * 24 * H * W), C8 * 8, H, W

    def x3_pack(self, x, out):
        if out.dtype == torch.float16:
            return self.h2_pack(x, out)
        xp, xbs, Cc, H, W = _view(x, "x3_pack.x")
        yp, ybs, c2, h2, w2 = self._x3view(out, "x3_pack.out")
        assert (Cc, H, W) == (c2, h2, w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("x3_pack",) + tuple(x.shape), lambda: self.lib.bfsr_x3_pack(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "x3_pack")
        return out

    def x3_unpack(self, x, out):
        if x.dtype == torch.float16:
            return self.h2_unpack(x, out)
        xp, xbs, Cc, H, W = self._x3view(x, "x3_unpack.x")
        yp, ybs, c2, h2, w2 = _view(out, "x3_unpack.out")
        assert (Cc, H, W) == (c2, h2, w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("x3_unpack",) + tuple(out.shape), lambda: self.lib.bfsr_x3_unpack(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "x3_unpack")
        return out

    def conv_x3s(self, x, pw, out, epi=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0, res2=None, alpha2=1.0, tune=0, y_fmt=0, up4=None):
        """3x3 conv over an x3 tensor `x` (weights: pack_conv_x3(w, 1)); `out` is an x3 view or an fp32 NCHW view; residuals
        are x3 views.  Same epilogue contract as conv().  h2 tensors (split == "f16x2") go to conv_h2x: same contract."""
        if x.dtype == torch.float16:
            return self.conv_h2x(x, pw, out, epi=epi, act=act, slope=slope, res1=res1, alpha1=alpha1, res2=res2, alpha2=alpha2, tune=tune, y_fmt=y_fmt, up4=up4)
        if y_fmt or up4 is not None:
            raise ValueError("conv_x3s: the quad-major output exists on the fp16-split kernels only")
        a = _lib.BfsrConvX3Args()
        a.x, a.x_bs, Cin, H, W = self._x3view(x, "conv_x3s.x")
        if out.dtype == torch.bfloat16:
            a.y, a.y_bs, Cout, H2, W2 = self._x3view(out, "conv_x3s.out")
            a.y_fmt = 1
        else:
            a.y, a.y_bs, Cout, H2, W2 = _view(out, "conv_x3s.out")
            a.y_fmt = 0
        if (Cin, Cout, H, W) != (pw.Cin, pw.Cout, H2, W2) or pw.KS != 3 or x.shape[0] != out.shape[0]:
            raise ValueError("conv_x3s: shape mismatch x%s out%s weight(Cout=%d,Cin=%d)" % (tuple(x.shape), tuple(out.shape), pw.Cout, pw.Cin))
        a.Cin, a.Cout = Cin, Cout
        a.w = pw.variant(1 if pw.arith == 0 else "bf16x3", lambda w_, m_: self._pack_raw_16(w_, 1, "bf16x3")).data_ptr()
        a.B, a.H, a.W = out.shape[0], H, W
        a.epi, a.act, a.slope, a.tune = _ptr(epi), act, slope, tune
        for name, t, al in (("res1", res1, alpha1), ("res2", res2, alpha2)):
            if t is not None:
                pp, bs, c, hh, ww = self._x3view(t, "conv_x3s." + name)
                assert (c, hh, ww) == (Cout, H, W)
                setattr(a, name, pp)
                setattr(a, name + "_bs", bs)
                setattr(a, "alpha" + name[-1], al)
        key = ("conv_x3s", Cin, Cout, out.shape[0], H, W, a.y_fmt)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv3x3_x3s(C.byref(a), self._stream())), "conv3x3_x3s")
        return out

    # ---- h2 tensors (activations stored in fp16: hi + lo planes) + the LDS-DMA fp16 conv over them (conv_h2s.hip) ---------------
    def h2_empty(self, B, Cc, H, W):
        """[B, C/8, 2, H, W, 8] fp16: x ~ hi + lo; channel slices `t[:, a//8:b//8]` are views."""
        if Cc % 8:
            raise ValueError("h2 tensors need a multiple of 8 channels")
        return torch.empty(B, Cc // 8, 2, H, W, 8, dtype=torch.float16, device=self.device)

    @staticmethod
    def _h2view(t, name="h2"):
        """(ptr, batch stride in fp16 elements, C, H, W) of an h2 view."""
        if t.dtype != torch.float16 or t.dim() != 6 or t.shape[2] != 2 or t.shape[5] != 8:
            raise ValueError("%s: need a [B,C/8,2,H,W,8] float16 tensor" % name)
        B, C8, _, H, W, _ = t.shape
        st = t.stride()
        if not (st[5] == 1 and st[4] == 8 and st[3] == 8 * W and st[2] == 8 * W * H and (C8 == 1 or st[1] == 16 * W * H)):
            raise ValueError("%s: not an h2 view (shape %s stride %s)" % (name, tuple(t.shape), st))
        return t.data_ptr(), (st[0] if B > 1 else C8 * 16 * H * W), C8 * 8, H, W

    def h2_pack(self, x, out):
        xp, xbs, Cc, H, W = _view(x, "h2_pack.x")
        yp, ybs, c2, h2, w2 = self._h2view(out, "h2_pack.out")
        assert (Cc, H, W) == (c2, h2, w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("h2_pack",) + tuple(x.shape), lambda: self.lib.bfsr_h2_pack(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self.range_flag.data_ptr(), self._stream())), "h2_pack")
        return out

    def h2_pack_pad(self, x, out):
        """h2_pack of x [B,Cs,H,W] into an h2 view with C >= Cs channels, the extra channels zero (= copying x into a zero-initialised [B,C,H,W] tensor and packing that)."""
        xp, xbs, Cs, H, W = _view(x, "h2_pack_pad.x")
        yp, ybs, c2, h2, w2 = self._h2view(out, "h2_pack_pad.out")
        assert Cs <= c2 and (H, W) == (h2, w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("h2_pack_pad", x.shape[0], Cs, c2, H, W), lambda: self.lib.bfsr_h2_pack_pad(xp, xbs, yp, ybs, x.shape[0], Cs, c2, H, W, self.range_flag.data_ptr(),
                                                                                                            self._stream())), "h2_pack_pad")
        return out

    def h2_unpack(self, x, out):
        xp, xbs, Cc, H, W = self._h2view(x, "h2_unpack.x")
        yp, ybs, c2, h2, w2 = _view(out, "h2_unpack.out")
        assert (Cc, H, W) == (c2, h2, w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("h2_unpack",) + tuple(out.shape), lambda: self.lib.bfsr_h2_unpack(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "h2_unpack")
        return out

    @staticmethod
    def pow2_scale(w):
        """The power of two that puts the largest |w| into [2^9, 2^10): fp16 hi/lo splits of w*scale keep their lo terms normal."""
        m = float(w.abs().max())
        if not (m > 0.0) or m != m or m == float("inf"):
            return 1.0
        import math
        return 2.0 ** (9 - math.floor(math.log2(m)))

    def _pack_h2x(self, w):
        Cout, Cin, KS, _ = w.shape
        if KS != 3 or Cin % 16:
            raise ValueError("conv_h2x: 3x3 weights with Cin % 16 == 0 only")
        scale = self.pow2_scale(w)
        mt = 1          # 64-cout workgroup tiles (round 3, BFSR_H2X_MT=2) were measured 5-8 % slower and are gone
        packed = torch.empty(self.lib.bfsr_conv_packed_size_h2x(Cout, Cin, mt), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv_weight_h2x(w.data_ptr(), Cout, Cin, mt, scale, packed.data_ptr()), "pack_h2x")
        return packed.to(self.device), scale, mt

    def conv_h2x(self, x, pw, out, epi=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0, res2=None, alpha2=1.0, tune=0, y_fmt=0, up4=None):
        """3x3 conv over an h2 tensor `x` at fp32-class accuracy (both planes x two-term fp16 weights, three products; conv_h2s.hip,
        conv3x3_h2x_kernel).  `pw` = pack_conv_x3(w, ...) (the fp16 packing is derived lazily from its kept OIHW copy); `out` is an
        h2 view (both planes) or an fp32 NCHW view; residuals are h2 views.  Same epilogue contract as conv()."""
        a = _lib.BfsrConvX3Args()
        a.x, a.x_bs, Cin, H, W = self._h2view(x, "conv_h2x.x")
        if out.dtype == torch.float16:
            a.y, a.y_bs, Cout, H2, W2 = self._h2view(out, "conv_h2x.out")
            a.y_fmt = 1
        else:
            a.y, a.y_bs, Cout, H2, W2 = _view(out, "conv_h2x.out")
            a.y_fmt = 2 if y_fmt else 0             # y_fmt=1: fp32 quad-major [B][Cout/4][H][W][4] in the same buffer
        if (Cin, Cout, H, W) != (pw.Cin, pw.Cout, H2, W2) or pw.KS != 3 or x.shape[0] != out.shape[0]:
            raise ValueError("conv_h2x: shape mismatch x%s out%s weight(Cout=%d,Cin=%d)" % (tuple(x.shape), tuple(out.shape), pw.Cout, pw.Cin))
        a.Cin, a.Cout = Cin, Cout
        wdata, scale, mt = pw.variant("h2x", lambda w_, m_: self._pack_h2x(w_))
        a.w, a.acc_scale, a.mtile, a.flag = wdata.data_ptr(), 1.0 / scale, mt, self.range_flag.data_ptr()
        a.B, a.H, a.W = out.shape[0], H, W
        a.epi, a.act, a.slope, a.tune = _ptr(epi), act, slope, tune
        for name, t, al in (("res1", res1, alpha1), ("res2", res2, alpha2)):
            if t is not None:
                pp, bs, c, hh, ww = self._h2view(t, "conv_h2x." + name)
                assert (c, hh, ww) == (Cout, H, W)
                setattr(a, name, pp)
                setattr(a, name + "_bs", bs)
                setattr(a, "alpha" + name[-1], al)
        if up4 is not None:                          # compact result of conv_up4_h2t(compact=True) at H/4 x W/4, added after the epilogue (quad-major output only)
            if a.y_fmt != 2 or up4.dtype != torch.float32 or tuple(up4.shape) != (out.shape[0], Cout * 9, H // 4, W // 4) or H % 4 or W % 4:
                raise ValueError("conv_h2x: up4 needs the quad-major output and a compact tensor [B, 9*Cout, H/4, W/4]")
            a.up4, a.up4_bs = up4.data_ptr(), (up4.stride(0) if up4.shape[0] > 1 else up4[0].numel())
            if not up4[0].is_contiguous():
                raise ValueError("conv_h2x: up4 must be contiguous per sample")
        key = ("conv_h2x", Cin, Cout, out.shape[0], H, W, a.y_fmt)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv3x3_h2x(C.byref(a), self._stream())), "conv3x3_h2x")
        return out

    # ---- a chain of h2x convs in one persistent launch (conv_chain.hip): the dense blocks of the RRDB encoder ---------------------
    def conv_chain(self, specs):
        """Prepare a chain of 3x3 convs over h2 tensors for ONE persistent launch (bfsr_conv_chain_*).  `specs` = list of dicts with the
        keyword arguments of conv_h2x: x, pw, out, epi, act, slope, res1, alpha1, res2, alpha2, y_fmt, plus `out2` (optional fp32 NCHW
        tensor that receives a second copy of the conv's result).  Every conv but the last must write an h2 view.  Returns a ConvChain;
        .run() launches it on the current stream; the results are bit-identical to conv_h2x launches of the same convs.  The chain holds pointers: the tensors it was built from must stay allocated."""
        n = len(specs)
        arr = (_lib.BfsrChainConv * n)()
        keep = []
        B = H = W = None
        for i, sp in enumerate(specs):
            a = arr[i]
            x, pw, out = sp["x"], sp["pw"], sp["out"]
            a.x, a.x_bs, Cin, h, w_ = self._h2view(x, "conv_chain.x")
            if out.dtype == torch.float16:
                a.y, a.y_bs, Cout, h2, w2 = self._h2view(out, "conv_chain.out")
                a.y_fmt = 1
            else:
                a.y, a.y_bs, Cout, h2, w2 = _view(out, "conv_chain.out")
                a.y_fmt = 2 if sp.get("y_fmt") else 0
            if (Cin, Cout, h, w_) != (pw.Cin, pw.Cout, h2, w2) or pw.KS != 3 or x.shape[0] != out.shape[0]:
                raise ValueError("conv_chain[%d]: shape mismatch x%s out%s weight(Cout=%d,Cin=%d)" % (i, tuple(x.shape), tuple(out.shape), pw.Cout, pw.Cin))
            if B is None:
                B, H, W = out.shape[0], h, w_
            elif (B, H, W) != (out.shape[0], h, w_):
                raise ValueError("conv_chain[%d]: every conv of a chain shares B, H, W" % i)
            a.Cin, a.Cout = Cin, Cout
            wdata, scale, _mt = pw.variant("h2x", lambda w__, m_: self._pack_h2x(w__))
            a.w, a.acc_scale = wdata.data_ptr(), 1.0 / scale
            epi = sp.get("epi")
            a.epi, a.act, a.slope = _ptr(epi), sp.get("act", ACT_NONE), sp.get("slope", 0.2)
            for name in ("res1", "res2"):
                t = sp.get(name)
                if t is not None:
                    pp, bs, c, hh, ww = self._h2view(t, "conv_chain." + name)
                    assert (c, hh, ww) == (Cout, H, W)
                    setattr(a, name, pp)
                    setattr(a, name + "_bs", bs)
                    setattr(a, "alpha" + name[-1], sp.get("alpha" + name[-1], 1.0))
            o2 = sp.get("out2")
            if o2 is not None:
                a.y2, a.y2_bs, c, hh, ww = _view(o2, "conv_chain.out2")
                assert (c, hh, ww) == (Cout, H, W) and o2.shape[0] == B
            keep.append((x, out, wdata, epi, sp.get("res1"), sp.get("res2"), o2))
        size = self.lib.bfsr_conv_chain_table_size(n)
        table = torch.zeros(size, dtype=torch.uint8)
        _lib.check(self.lib.bfsr_conv_chain_prepare(arr, n, B, H, W, table.data_ptr()), "conv_chain_prepare")
        words = self.lib.bfsr_conv_chain_progress_words(table.data_ptr())
        return ConvChain(self, table, table.to(self.device), torch.zeros(words, dtype=torch.int32, device=self.device), keep,
                         ("conv_chain", n, sum(sp["pw"].Cin * sp["pw"].Cout for sp in specs), B, H, W))      # key: convs, sum of Cin x Cout (-> flops), shape

    def h2_pack_s2d(self, x, out):
        """fp32 [B,C,2h,2w] view -> h2 view with 4C channels at h x w (space to depth): channel q*C + c = pixels (2y+qy, 2x+qx) of channel c,
        q = qy*2 + qx -- the form in which channels at output resolution enter conv_up2_h2t as key chunks."""
        xp, xbs, Cc, H, W = _view(x, "h2_pack_s2d.x")
        yp, ybs, c2, h2, w2 = self._h2view(out, "h2_pack_s2d.out")
        assert (4 * Cc, H, W) == (c2, 2 * h2, 2 * w2) and x.shape[0] == out.shape[0]
        _lib.check(self._launch(("h2_pack_s2d",) + tuple(x.shape), lambda: self.lib.bfsr_h2_pack_s2d(xp, xbs, yp, ybs, x.shape[0], Cc, h2, w2, self.range_flag.data_ptr(), self._stream())),
                   "h2_pack_s2d")
        return out

    def pack_conv_up2_h2t(self, w_taps, w_key=None):
        """OIHW 3x3 weights of the conv over cat([key at output resolution, nearest-x2-upsampled taps]) -> conv_up2_h2t's packing.  w_taps
        [Cout,Ct,3,3]: per output parity the window taps that fall on one source pixel are summed (in double); w_key [Cout,Ck,3,3] (optional):
        single window taps per space-to-depth plane.  Everything is scaled by one power of two and split into fp16 hi + lo."""
        w = w_taps.detach().to("cpu", torch.float32).contiguous()
        Cout, Ct = w.shape[0], w.shape[1]
        wk = None if w_key is None else w_key.detach().to("cpu", torch.float32).contiguous()
        Ck = 0 if wk is None else wk.shape[1]
        if Cout % 32 or Ct % 16 or Ck % 16 or tuple(w.shape[2:]) != (3, 3) or (wk is not None and (wk.shape[0] != Cout or tuple(wk.shape[2:]) != (3, 3))):
            raise ValueError("pack_conv_up2_h2t: unsupported shapes %s %s" % (tuple(w.shape), None if wk is None else tuple(wk.shape)))
        wd, sets = w.double(), ((0,), (1, 2), (0, 1), (2,))
        m = max(float(wd[:, :, list(r)][:, :, :, list(c)].sum((2, 3)).abs().max()) for r in sets for c in sets)
        if wk is not None:
            m = max(m, float(wk.abs().max()))
        scale = self.pow2_scale(torch.tensor([m]))
        packed = torch.empty(self.lib.bfsr_conv_up2_h2t_packed_size(Cout, Ct, Ck), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv_up2_h2t(w.data_ptr(), 0 if wk is None else wk.data_ptr(), Cout, Ct, Ck, scale, packed.data_ptr()), "pack_conv_up2_h2t")
        return packed.to(self.device), 1.0 / scale, Cout, Ct, Ck

    def conv_up2_h2t(self, x, packed, out, pre_add=None):
        """conv3x3(cat([key, nearest_up2(taps)])) at source resolution (conv_up2_h2t.hip: parity-decomposed, two-term fp16 split, three products).
        `x`: h2 tensor [B,(Ct + 4*Ck)/8,2,h,w,8] = the taps followed by the space-to-depth planes of the key channels (h2_pack_s2d); `out` and
        `pre_add` (may be `out`) are fp32 buffers of shape [B,Cout,2h,2w] holding the QUAD-MAJOR layout [B][Cout/4][2h][2w][4]: out = conv + pre_add."""
        wts, acc_scale, Cout, Ct, Ck = packed
        a = _lib.BfsrUp2H2Args()
        a.x, a.x_bs, cin, h, w = self._h2view(x, "conv_up2_h2t.x")
        a.y, a.y_bs, co, H, W = _view(out, "conv_up2_h2t.out")
        if (cin, co, 2 * h, 2 * w) != (Ct + 4 * Ck, Cout, H, W) or x.shape[0] != out.shape[0]:
            raise ValueError("conv_up2_h2t: shape mismatch x%s out%s weight(Cout=%d,Ct=%d,Ck=%d)" % (tuple(x.shape), tuple(out.shape), Cout, Ct, Ck))
        a.Cin, a.Ckey, a.Cout, a.y_fmt = cin, Ck, Cout, 1
        a.w, a.acc_scale = wts.data_ptr(), acc_scale
        a.B, a.h, a.w_ = out.shape[0], h, w
        if pre_add is not None:
            a.pre_add, a.pre_add_bs, c, hh, ww = _view(pre_add, "conv_up2_h2t.pre_add")
            assert (c, hh, ww) == (Cout, H, W)
        key = ("conv_up2_h2t", Ct, Ck, Cout, out.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up2_h2t(C.byref(a), self._stream())), "conv2d_up2_h2t")
        return out

    def pack_conv_up4_h2t(self, w_taps):
        """OIHW 3x3 weights over the nearest-x4-upsampled tap channels -> conv_up4_h2t's packing (25 pre-summed blocks per 16-channel chunk)."""
        w = w_taps.detach().to("cpu", torch.float32).contiguous()
        Cout, Ct = w.shape[0], w.shape[1]
        if w.shape[2:] != (3, 3) or Cout % 32 or Ct % 16:
            raise ValueError("pack_conv_up4_h2t: unsupported shape %s" % (tuple(w.shape),))
        wd, sets = w.double(), ((0,), (1, 2), (0, 1, 2), (0, 1), (2,))
        m = max(float(wd[:, :, list(r)][:, :, :, list(c)].sum((2, 3)).abs().max()) for r in sets for c in sets)
        scale = self.pow2_scale(torch.tensor([m]))          # the largest pre-summed block entry into [2^9, 2^10)
        packed = torch.empty(self.lib.bfsr_conv_up4_h2t_packed_size(Cout, Ct), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv_up4_h2t(w.data_ptr(), Cout, Ct, scale, packed.data_ptr()), "pack_conv_up4_h2t")
        return packed.to(self.device), 1.0 / scale, Cout, Ct

    def conv_up4_h2t(self, x, packed, out, pre_add=None, compact=False):
        """conv3x3(nearest_up4(taps)) + pre_add at source resolution (conv_up4_h2t.hip: phase-decomposed, two-term fp16 split, three products).
        `x`: h2 tensor [B,Ct/8,2,h,w,8]; `out` and `pre_add` (may be `out`) are fp32 buffers of shape [B,Cout,4h,4w] holding the QUAD-MAJOR layout.
        compact=True: `out` is a [B, 9*Cout, h, w] fp32 buffer that receives the nine phase-class values per source pixel and channel quad
        ([Cout/4][h][9][w][4]; no pre_add) -- what conv_h2x(up4=...) adds to the conv over the channels at output resolution."""
        wts, acc_scale, Cout, Ct = packed
        a = _lib.BfsrUp2H2Args()
        a.x, a.x_bs, cin, h, w = self._h2view(x, "conv_up4_h2t.x")
        a.y, a.y_bs, co, H, W = _view(out, "conv_up4_h2t.out")
        if compact:
            if pre_add is not None or (cin, co, H, W) != (Ct, 9 * Cout, h, w) or x.shape[0] != out.shape[0] or not out[0].is_contiguous():
                raise ValueError("conv_up4_h2t(compact): out must be a contiguous [B, 9*Cout, h, w] buffer and there is no pre_add")
            a.Cin, a.Ckey, a.Cout, a.y_fmt = cin, 0, Cout, 3
            a.w, a.acc_scale = wts.data_ptr(), acc_scale
            a.B, a.h, a.w_ = out.shape[0], h, w
            key = ("conv_up4_h2t", Ct, 9, Cout, out.shape[0], 4 * h, 4 * w)
            _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up4_h2t(C.byref(a), self._stream())), "conv2d_up4_h2t")
            return out
        if (cin, co, 4 * h, 4 * w) != (Ct, Cout, H, W) or x.shape[0] != out.shape[0]:
            raise ValueError("conv_up4_h2t: shape mismatch x%s out%s weight(Cout=%d,Ct=%d)" % (tuple(x.shape), tuple(out.shape), Cout, Ct))
        a.Cin, a.Ckey, a.Cout, a.y_fmt = cin, 0, Cout, 1
        a.w, a.acc_scale = wts.data_ptr(), acc_scale
        a.B, a.h, a.w_ = out.shape[0], h, w
        if pre_add is not None:
            a.pre_add, a.pre_add_bs, c, hh, ww = _view(pre_add, "conv_up4_h2t.pre_add")
            assert (c, hh, ww) == (Cout, H, W)
        key = ("conv_up4_h2t", Ct, 0, Cout, out.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d_up4_h2t(C.byref(a), self._stream())), "conv2d_up4_h2t")
        return out

    def pack_conv_h2s(self, w, mtile=None):
        """OIHW 3x3 fp32 weights -> fp16 packing of conv_h2s (32- or 64-cout workgroup tiles, 16-channel chunks)."""
        w = w.detach().to("cpu", torch.float32).contiguous()
        Cout, Cin, KS, _ = w.shape
        if KS != 3 or Cin % 32:
            raise ValueError("conv_h2s: 3x3 weights with Cin % 32 == 0 only")
        # round 6: 64-cout workgroup tiles where the conv has them (conv5 of a dense block, trunk convs, coef | freq, the priors): the input tile is
        # staged once per 64 output channels.  Same summation order per output element -> the bits of the 32-cout form (BFSR_H2S_MT=1 keeps that form).
        mt = mtile if mtile is not None else (2 if (Cout % 64 == 0 and os.environ.get("BFSR_H2S_MT", "2") != "1") else 1)
        packed = torch.empty(self.lib.bfsr_conv_packed_size_h2s_mt(Cout, Cin, mt), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_conv_weight_h2s_mt(w.data_ptr(), Cout, Cin, mt, packed.data_ptr()), "pack_h2s")
        return PackedConv(packed.to(self.device), Cout, Cin, 3, mt, fixed=True, w=None, ops=self)

    def conv_h2s(self, x, pw, out, epi=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0, res2=None, alpha2=1.0, hi_only=False, tune=0):
        """3x3 conv over an h2 tensor `x` (weights: pack_conv_h2s); `out` is an h2 view (hi_only: the lo plane is not written --
        for outputs that only feed convs) or an fp32 NCHW view; residuals are h2 views.  Same epilogue contract as conv()."""
        a = _lib.BfsrConvX3Args()
        a.x, a.x_bs, Cin, H, W = self._h2view(x, "conv_h2s.x")
        if out.dtype == torch.float16:
            a.y, a.y_bs, Cout, H2, W2 = self._h2view(out, "conv_h2s.out")
            a.y_fmt = 2 if hi_only else 1
        else:
            a.y, a.y_bs, Cout, H2, W2 = _view(out, "conv_h2s.out")
            a.y_fmt = 0
        if (Cin, Cout, H, W) != (pw.Cin, pw.Cout, H2, W2) or pw.KS != 3 or x.shape[0] != out.shape[0]:
            raise ValueError("conv_h2s: shape mismatch x%s out%s weight(Cout=%d,Cin=%d)" % (tuple(x.shape), tuple(out.shape), pw.Cout, pw.Cin))
        a.Cin, a.Cout = Cin, Cout
        # bit 8: keep the weights streamed.  The LDS-resident form (round 6) is bit-identical and NOT faster (config 5: 141.71 vs 141.67 ms, profiles/r06o_*):
        # off unless BFSR_H2S_RES=1
        a.w, a.mtile = pw.data.data_ptr(), pw.mtile | (0 if os.environ.get("BFSR_H2S_RES", "0") == "1" else 0x100)
        a.B, a.H, a.W = out.shape[0], H, W
        a.epi, a.act, a.slope, a.tune = _ptr(epi), act, slope, tune
        for name, t, al in (("res1", res1, alpha1), ("res2", res2, alpha2)):
            if t is not None:
                pp, bs, c, hh, ww = self._h2view(t, "conv_h2s." + name)
                assert (c, hh, ww) == (Cout, H, W)
                setattr(a, name, pp)
                setattr(a, name + "_bs", bs)
                setattr(a, "alpha" + name[-1], al)
        key = ("conv_h2s", Cin, Cout, out.shape[0], H, W, a.y_fmt, pw.mtile)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv3x3_h2s(C.byref(a), self._stream())), "conv3x3_h2s")
        return out

    def vec(self, t):
        """A per-channel parameter vector on the device."""
        return t.detach().reshape(-1).to(device=self.device, dtype=torch.float32).contiguous()

    def pack_epilogue(self, Cout, bias=None, aff_shift=None, aff_scale=None, aff_post=None, post_scale=None):
        """Per-channel epilogue vectors -> the packed [Cout,8] device tensor of the C ABI (None if all neutral)."""
        vs = (bias, aff_shift, aff_scale, aff_post, post_scale)
        if all(v is None for v in vs):
            return None
        e = torch.zeros(Cout, 8, dtype=torch.float32)
        e[:, 2] = 1.0
        e[:, 4] = 1.0
        for col, v in enumerate(vs):
            if v is not None:
                e[:, col] = v.detach().reshape(-1).to("cpu", torch.float32)
        return e.to(self.device)

    def conv(self, x, pw, out, in_shift=0, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0,
             res2=None, alpha2=1.0, tune=0, stage2=None, bias=None, aff_shift=None, aff_scale=None, aff_post=None,
             post_scale=None):
        """epi: packed per-channel parameters from pack_epilogue (the loose vector kwargs are packed on the fly, for tests).
        stage2 = (PackedConv 1x1, epi2, act2): fused second conv, `out` then has stage2 Cout channels."""
        if epi is None and any(v is not None for v in (bias, aff_shift, aff_scale, aff_post, post_scale)):
            epi = self.pack_epilogue(pw.Cout, bias, aff_shift, aff_scale, aff_post, post_scale)
        xp, xbs, Cin, Hs, Ws = _view(x, "conv.x")
        yp, ybs, Cout, H, W = _view(out, "conv.out")
        c_final = Cout
        if stage2 is not None:
            pw2, epi2, act2 = stage2
            if pw2.KS != 1 or pw2.mtile != 2 or pw2.Cin != pw.Cout or pw2.Cout != Cout:
                raise ValueError("conv: bad fused stage")
            Cout = pw.Cout
        if Cin != pw.Cin or Cout != pw.Cout or (Hs << in_shift) != H or (Ws << in_shift) != W or x.shape[0] != out.shape[0]:
            raise ValueError("conv: shape mismatch x%s out%s weight(Cout=%d,Cin=%d) in_shift=%d" %
                             (tuple(x.shape), tuple(out.shape), pw.Cout, pw.Cin, in_shift))
        mtile, wdata = pw.mtile, pw.data
        if not pw.fixed and stage2 is None and mtile == 2 and pw.KS == 3:
            # small grids: one 32-wide cout tile per workgroup doubles the workgroup count (measured on MI355X:
            # RDB conv5 192->64 @ 8x160x160 goes from 99 to 120 TFLOP/s)
            tiles = ((W + 31) // 32) * ((H + 7) // 8) * out.shape[0]
            if tiles * ((Cout + 63) // 64) < 1280:
                mtile, wdata = 1, pw.variant(1)
        a = _lib.BfsrConvArgs()
        a.x, a.x_bs, a.Cin = xp, xbs, Cin
        a.w = wdata.data_ptr()
        a.y, a.y_bs, a.Cout = yp, ybs, Cout
        if stage2 is not None:
            a.w2, a.C2, a.epi2, a.act2 = pw2.data.data_ptr(), c_final, _ptr(epi2), act2
        a.B, a.H, a.W, a.KS, a.in_shift, a.mtile = out.shape[0], H, W, pw.KS, in_shift, mtile
        a.epi = _ptr(epi)
        if pre_add is not None:
            p, bs, c, h, w = _view(pre_add, "conv.pre_add")
            assert (c, h, w) == (Cout, H, W)
            a.pre_add, a.pre_add_bs = p, bs
        a.act, a.slope, a.tune = act, slope, tune
        if res1 is not None:
            p, bs, c, h, w = _view(res1, "conv.res1")
            assert (c, h, w) == (Cout, H, W)
            a.res1, a.res1_bs, a.alpha1 = p, bs, alpha1
        if res2 is not None:
            p, bs, c, h, w = _view(res2, "conv.res2")
            assert (c, h, w) == (Cout, H, W)
            a.res2, a.res2_bs, a.alpha2 = p, bs, alpha2
        key = ("conv", pw.KS, mtile, Cin, Cout, out.shape[0], H, W) if stage2 is None else ("conv+1x1", Cin, Cout, out.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv2d(C.byref(a), self._stream())), "conv2d")
        return out

    # ---- flow pointwise ---------------------------------------------------------------------
    def flow_pointwise(self, z_in, z_out, reverse, h_aff=None, h_ft=None, w=None, an_bias=None, an_escale=None,
                       eps=1e-4, wt=None):
        a = _lib.BfsrFlowArgs()
        a.z_in, a.z_in_bs, Cc, H, W = _view(z_in, "flow.z_in")
        a.z_out, a.z_out_bs, c2, h2, w2 = _view(z_out, "flow.z_out")
        assert (Cc, H, W) == (c2, h2, w2)
        if h_aff is not None:
            a.h_aff, a.h_aff_bs, c, h, ww = _view(h_aff, "flow.h_aff")
            assert (c, h, ww) == (2 * (Cc - Cc // 2), H, W)
        if h_ft is not None:
            a.h_ft, a.h_ft_bs, c, h, ww = _view(h_ft, "flow.h_ft")
            assert (c, h, ww) == (2 * Cc, H, W)
        a.w, a.wt, a.an_bias, a.an_escale = _ptr(w), _ptr(wt), _ptr(an_bias), _ptr(an_escale)
        a.B, a.C, a.H, a.W = z_in.shape[0], Cc, H, W
        a.reverse, a.eps = int(bool(reverse)), eps
        key = ("flow", int(bool(reverse)), Cc, z_in.shape[0], H, W, h_aff is not None, h_ft is not None, w is not None)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_flow_pointwise(C.byref(a), self._stream())),
                   "flow_pointwise(C=%d)" % Cc)
        return z_out

    # ---- the sequential part of a coupled FlowStep in two kernels (coupling.hip, conv_h2s.hip) ---------------------------------
    def pack_coupling_head(self, w0_z1, w2, shift0, scale0, shift2, scale2):
        """fAffine.0 restricted to the z1 rows [64,Cz,3,3] + fAffine.2 [64,64(,1,1)] as two-term fp16 splits of w * 2^k, and their
        ActNorm (bias, exp(logs)) vectors."""
        w2 = w2.detach().to("cpu", torch.float32).reshape(64, 64).contiguous()
        if w0_z1 is None:                   # no 3x3 stage: hid = relu(AN2(W2 . relu(AN0(pre_aff)))) (the hoisted fFeatures nets)
            w0, Cz, s0 = None, 0, 1.0
        else:
            w0 = w0_z1.detach().to("cpu", torch.float32).contiguous()
            Cz, s0 = w0.shape[1], self.pow2_scale(w0)
            if w0.shape[0] != 64:
                raise ValueError("pack_coupling_head: unsupported shape %s" % (tuple(w0.shape),))
        n = self.lib.bfsr_coupling_head_packed_size(Cz)
        if n <= 0:
            raise ValueError("pack_coupling_head: unsupported z1 width %d" % Cz)
        s2 = self.pow2_scale(w2)
        packed = torch.empty(n, dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_coupling_head(w0.data_ptr() if w0 is not None else None, w2.data_ptr(), Cz, s0, s2, packed.data_ptr()), "pack_coupling_head")

        def epi(shift, scale):
            e = torch.zeros(64, 4, dtype=torch.float32)
            e[:, 0] = shift.detach().reshape(-1).to("cpu", torch.float32)
            e[:, 1] = scale.detach().reshape(-1).to("cpu", torch.float32)
            return e.to(self.device)
        return packed.to(self.device), epi(shift0, scale0), epi(shift2, scale2), Cz, 1.0 / s0, 1.0 / s2

    def coupling_head(self, z, packed, pre_aff, hid, pre_fmt=0):
        """hid (h2 tensor [B,8,2,H,W,8]) = relu(AN2(W2 . relu(AN0(conv3x3(z[:, :Cz]) + pre_aff)))) on the two-term fp16 split, the 1x1
        chained in registers.  pre_fmt=1: pre_aff is handed over quad-major ([B][16][H][W][4] in the same buffer)."""
        wts, e0, e2, Cz, as0, as2 = packed
        a = _lib.BfsrCouplingHeadArgs()
        a.pre_aff, a.pre_aff_bs, c1, H, W = _view(pre_aff, "coupling_head.pre_aff")
        if Cz:
            a.z, a.z_bs, Cc, h0, w0_ = _view(z, "coupling_head.z")
            assert Cc >= Cz and (h0, w0_) == (H, W)
        a.hid, a.hid_bs, c2, h2, w2 = self._h2view(hid, "coupling_head.hid")
        assert c1 == 64 and (c2, h2, w2) == (64, H, W)
        a.Cz, a.w, a.epi0, a.epi2 = Cz, wts.data_ptr(), e0.data_ptr(), e2.data_ptr()
        a.acc_scale0, a.acc_scale2, a.pre_fmt = as0, as2, int(pre_fmt)
        a.B, a.H, a.W, a.flag = pre_aff.shape[0], H, W, self.range_flag.data_ptr()
        key = ("coupling_head", Cz, pre_aff.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_coupling_head(C.byref(a), self._stream())), "coupling_head")
        return hid

    def pack_coupling_tail(self, w4, bias, post_scale):
        """fAffine.4 (Conv2dZeros) [Cout,64,3,3] in conv_h2x's packing (two-term fp16 split of w * 2^k) + its bias and exp(3*logs)."""
        w = w4.detach().to("cpu", torch.float32).contiguous()
        Cout, Cin = w.shape[0], w.shape[1]
        if Cin != 64 or Cout > 32 or tuple(w.shape[2:]) != (3, 3):
            raise ValueError("pack_coupling_tail: unsupported shape %s" % (tuple(w.shape),))
        scale = self.pow2_scale(w)
        packed = torch.empty(self.lib.bfsr_coupling_tail_packed_size(Cin, Cout), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_coupling_tail(w.data_ptr(), Cin, Cout, scale, packed.data_ptr()), "pack_coupling_tail")
        return packed.to(self.device), self.vec(bias), self.vec(post_scale), Cout, 1.0 / scale

    def coupling_tail(self, hid, packed, z_in, z_out, reverse, h_ft=None, w=None, an_bias=None, an_escale=None, eps=1e-4, h_ft_fmt=0):
        """h_aff = Conv2dZeros(hid) over the h2 tensor `hid`, then the pointwise chain of flow_pointwise(z_in, z_out, reverse,
        h_aff=h_aff, h_ft, w, an_*) as the conv's epilogue (conv3x3_h2x_kernel, coupling-tail epilogue)."""
        wts, bias, ps, Cout, acc_scale = packed
        a = _lib.BfsrCouplingTailArgs()
        a.hid, a.hid_bs, a.Cin, H, W = self._h2view(hid, "coupling_tail.hid")
        a.z_in, a.z_in_bs, Cc, h1, w1 = _view(z_in, "coupling_tail.z_in")
        a.z_out, a.z_out_bs, c2, h2, w2 = _view(z_out, "coupling_tail.z_out")
        assert (Cc, h1, w1) == (c2, h2, w2) == (Cc, H, W) and Cout == 2 * (Cc - Cc // 2)
        if h_ft is not None:
            a.h_ft, a.h_ft_bs, c, h, ww = _view(h_ft, "coupling_tail.h_ft")
            assert (c, h, ww) == (2 * Cc, H, W)
        a.w, a.acc_scale, a.bias, a.post_scale = wts.data_ptr(), acc_scale, bias.data_ptr(), ps.data_ptr()
        a.wmat, a.an_bias, a.an_escale = _ptr(w), _ptr(an_bias), _ptr(an_escale)
        a.B, a.C, a.H, a.W, a.reverse, a.eps, a.h_ft_fmt = z_in.shape[0], Cc, H, W, int(bool(reverse)), eps, int(h_ft_fmt)
        a.flag = self.range_flag.data_ptr()
        key = ("coupling_tail", int(bool(reverse)), Cc, z_in.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_coupling_tail(C.byref(a), self._stream())), "coupling_tail(C=%d)" % Cc)
        return z_out

    # ---- the coupled FlowStep of the wide level (C = 96) as two streaming kernels (coupling_wide.hip, round 6) ----------------------
    def pack_coupling_wide(self, w0_z1, w2, shift0, scale0, shift2, scale2, w4, bias4, post4):
        """fAffine.0 restricted to the z1 rows [64,Cz,3,3], fAffine.2 [64,64(,1,1)] with their ActNorm (bias, exp(logs)) vectors, and fAffine.4
        (Conv2dZeros [96,64,3,3], bias, exp(3*logs)): two-term fp16 splits of w * 2^k in the layouts the two kernels stream / keep resident."""
        w0 = w0_z1.detach().to("cpu", torch.float32).contiguous()
        w2 = w2.detach().to("cpu", torch.float32).reshape(64, 64).contiguous()
        w4 = w4.detach().to("cpu", torch.float32).contiguous()
        Cz = w0.shape[1]
        if w0.shape[0] != 64 or Cz % 16 or tuple(w4.shape) != (96, 64, 3, 3):
            raise ValueError("pack_coupling_wide: unsupported shapes %s %s" % (tuple(w0.shape), tuple(w4.shape)))
        s0, s2, s4 = self.pow2_scale(w0), self.pow2_scale(w2), self.pow2_scale(w4)
        p0 = torch.empty(self.lib.bfsr_coupling_wide_conv_packed_size(64, Cz), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_coupling_wide_conv(w0.data_ptr(), 64, Cz, s0, p0.data_ptr()), "pack_coupling_wide_conv(0)")
        p2 = torch.empty(self.lib.bfsr_coupling_wide_w2_packed_size(), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_coupling_wide_w2(w2.data_ptr(), s2, p2.data_ptr()), "pack_coupling_wide_w2")
        p4 = torch.empty(self.lib.bfsr_coupling_wide_conv_packed_size(96, 64), dtype=torch.int16)
        _lib.check(self.lib.bfsr_pack_coupling_wide_conv(w4.data_ptr(), 96, 64, s4, p4.data_ptr()), "pack_coupling_wide_conv(4)")

        def epi(shift, scale):
            e = torch.zeros(64, 4, dtype=torch.float32)
            e[:, 0] = shift.detach().reshape(-1).to("cpu", torch.float32)
            e[:, 1] = scale.detach().reshape(-1).to("cpu", torch.float32)
            return e.to(self.device)
        return dict(Cz=Cz, w0=p0.to(self.device), as0=1.0 / s0, w2=p2.to(self.device), as2=1.0 / s2, e0=epi(shift0, scale0), e2=epi(shift2, scale2),
                    w4=p4.to(self.device), as4=1.0 / s4, bias=self.vec(bias4), ps=self.vec(post4))

    def pack_wide_wmat(self, w):
        """A [96,96] matrix of the invertible 1x1 conv with its K axis in the order coupling_wide_tail contracts it."""
        w = w.detach().to("cpu", torch.float32).contiguous()
        if tuple(w.shape) != (96, 96):
            raise ValueError("pack_wide_wmat: [96, 96] only")
        out = torch.empty(96 * 96, dtype=torch.float32)
        _lib.check(self.lib.bfsr_pack_coupling_wide_wmat(w.data_ptr(), out.data_ptr()), "pack_coupling_wide_wmat")
        return out.to(self.device)

    def coupling_wide_head(self, z1h, packed, pre_h2, hid):
        """hid (h2, 64 ch) = relu(AN2(W2 . relu(AN0(conv3x3(z1h) + pre_h2)))): z1h = h2 view of the step's z1 channels, pre_h2 = h2 view (64 ch)."""
        a = _lib.BfsrWideHeadArgs()
        a.z1, a.z1_bs, Cz, H, W = self._h2view(z1h, "coupling_wide_head.z1h")
        a.pre, a.pre_bs, c1, h1, w1 = self._h2view(pre_h2, "coupling_wide_head.pre")
        a.hid, a.hid_bs, c2, h2, w2 = self._h2view(hid, "coupling_wide_head.hid")
        if Cz != packed["Cz"] or (c1, h1, w1) != (64, H, W) or (c2, h2, w2) != (64, H, W):
            raise ValueError("coupling_wide_head: shape mismatch z1h%s pre%s hid%s" % (tuple(z1h.shape), tuple(pre_h2.shape), tuple(hid.shape)))
        a.Cz, a.w0, a.acc_scale0, a.w2, a.acc_scale2 = Cz, packed["w0"].data_ptr(), packed["as0"], packed["w2"].data_ptr(), packed["as2"]
        a.epi0, a.epi2 = packed["e0"].data_ptr(), packed["e2"].data_ptr()
        a.B, a.H, a.W, a.flag = z1h.shape[0], H, W, self.range_flag.data_ptr()
        key = ("coupling_wide_head", Cz, z1h.shape[0], H, W)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_coupling_wide_head(C.byref(a), self._stream())), "coupling_wide_head")
        return hid

    def coupling_wide_tail(self, hid, packed, z_in, z_out, reverse, h_ft=None, w=None, an_bias=None, an_escale=None, eps=1e-4, h_ft_fmt=0, z1h=None):
        """h_aff = Conv2dZeros(hid) (64 -> 96), then flow_pointwise(z_in, z_out, reverse, h_aff, h_ft, w, an_*) as the conv's epilogue;
        `w` = pack_wide_wmat(W) (None: no matvec, forward only); z1h (optional h2 view, 48 ch) receives the first 48 result channels."""
        a = _lib.BfsrWideTailArgs()
        a.hid, a.hid_bs, cin, H, W = self._h2view(hid, "coupling_wide_tail.hid")
        a.z_in, a.z_in_bs, Cc, h1, w1 = _view(z_in, "coupling_wide_tail.z_in")
        a.z_out, a.z_out_bs, c2, h2, w2 = _view(z_out, "coupling_wide_tail.z_out")
        assert cin == 64 and (Cc, h1, w1) == (c2, h2, w2) == (96, H, W)
        if h_ft is not None:
            a.h_ft, a.h_ft_bs, c, h, ww = _view(h_ft, "coupling_wide_tail.h_ft")
            assert (c, h, ww) == (2 * Cc, H, W)
        if z1h is not None:
            a.z1h, a.z1h_bs, c, h, ww = self._h2view(z1h, "coupling_wide_tail.z1h")
            assert (c, h, ww) == (48, H, W)
        a.w, a.acc_scale, a.bias, a.post_scale = packed["w4"].data_ptr(), packed["as4"], packed["bias"].data_ptr(), packed["ps"].data_ptr()
        a.wperm, a.an_bias, a.an_escale = _ptr(w), _ptr(an_bias), _ptr(an_escale)
        a.B, a.C, a.H, a.W, a.reverse, a.eps, a.h_ft_fmt = z_in.shape[0], Cc, H, W, int(bool(reverse)), eps, int(h_ft_fmt)
        a.flag = self.range_flag.data_ptr()
        key = ("coupling_wide_tail", int(bool(reverse)), z_in.shape[0], H, W, h_ft is not None, w is not None, z1h is not None)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_coupling_wide_tail(C.byref(a), self._stream())), "coupling_wide_tail")
        return z_out

    def conv_h2r(self, x, packed, out, epi=None, act=ACT_NONE, slope=0.2, y_fmt=0):
        """3x3 conv 64 -> Cout <= 32 over the h2 tensor `x` on the coupling tail's conv kernel (coupling_tail.hip, plain epilogue);
        `packed` = pack_coupling_tail(w, ...) (only its weights and scale are used here); out fp32 NCHW or, y_fmt=1, quad-major."""
        wts, _b, _ps, Cout, acc_scale = packed
        a = _lib.BfsrConvX3Args()
        a.x, a.x_bs, Cin, H, W = self._h2view(x, "conv_h2r.x")
        a.y, a.y_bs, co, H2, W2 = _view(out, "conv_h2r.out")
        if (Cin, co, H, W) != (64, Cout, H2, W2) or x.shape[0] != out.shape[0]:
            raise ValueError("conv_h2r: shape mismatch x%s out%s Cout=%d" % (tuple(x.shape), tuple(out.shape), Cout))
        a.Cin, a.Cout, a.w, a.acc_scale = 64, Cout, wts.data_ptr(), acc_scale
        a.B, a.H, a.W, a.y_fmt = out.shape[0], H, W, 2 if y_fmt else 0
        a.epi, a.act, a.slope = _ptr(epi), act, slope
        key = ("conv_h2r", 64, Cout, out.shape[0], H, W, a.y_fmt)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_conv3x3_h2r(C.byref(a), self._stream())), "conv3x3_h2r")
        return out

    # ---- range guard of the fp16 split ------------------------------------------------------------------------------------------
    def check_range(self):
        """Raise if a kernel of the two-term fp16 split met a value it cannot represent (|x| >= 2^15, inf or NaN) since the last
        call.  One 4-byte device->host copy (synchronises the current stream); the engines call it once per pass."""
        v = int(self.range_flag.item())
        if v:
            self.range_flag.zero_()
            if v & 4:
                raise RuntimeError("bfsr_amd: a dependency wait of the fused conv chain timed out (flag 0x%x): results are invalid" % v)
            raise RuntimeError("bfsr_amd: a value left the range of the two-term fp16 split (flag 0x%x: bit 0 = operand >= 65504 or NaN, "
                               "bit 1 = non-finite flow state, bit 3 = a channel that is tiny everywhere); the guarded entry points re-run such "
                               "a pass under BFSR_SPLIT=bf16x3 (fp32's exponent range, six products) automatically" % v)

    def check_channels(self, x, gain=None, tiny=2.0 ** -7, huge=65504.0, ratio=2.0 ** -5):
        """Per-sample, per-channel dynamic-range check of an fp32 tensor that enters a region computed with the fp16-pair split (range_check.hip):
        raises bit 0 of the range flag when a channel reaches `huge`, bit 3 when a channel of a sample is tiny (0 < max |x| < tiny) AND the absolute
        error the pair leaves on it (2^-25 x the consumer's weight mass on that channel) exceeds 2^-20 of the largest per-channel contribution of the
        same sample: max_c(m_c g_c) < g_c * ratio.  `gain` = channel_gain(...) of the convs that read x ([C] device floats in [0, 1]); None = 1 for
        every channel, i.e. the rule fires only when the whole sample is tiny.  tiny=0 checks the overflow side only.  Asynchronous; the scratch is
        private to the current stream (the two half-batch lanes of the RRDB check concurrently); guard.run_guarded reads the flag at the end of the
        pass.  No-op under the bf16x3 split and the native fp32 MFMA."""
        if self.conv_mode != "x3" or self.split != "f16x2":
            return
        xp, xbs, Cc, H, W = _view(x, "check_channels.x")
        if gain is not None and (gain.dtype != torch.float32 or gain.numel() != Cc or not gain.is_contiguous() or gain.device != self.device):
            raise ValueError("check_channels: gain must be a contiguous fp32 device vector of %d channels" % Cc)
        n = self.lib.bfsr_channel_range_scratch(x.shape[0], Cc)
        if self._range_scratch is None:
            self._range_scratch = {}
        sk = torch.cuda.current_stream(self.device).cuda_stream
        sc = self._range_scratch.get(sk)
        if sc is None or sc.numel() < n:
            sc = self._range_scratch[sk] = torch.empty(n, dtype=torch.float32, device=self.device)    # allocated on (and only ever used on) this stream
        _lib.check(self._launch(("range_check", Cc, x.shape[0], H, W), lambda: self.lib.bfsr_channel_range_check(
            xp, xbs, x.shape[0], Cc, H, W, tiny, huge, ratio, _ptr(gain), sc.data_ptr(), self.range_flag.data_ptr(), self._stream())), "channel_range_check")

    def channel_gain(self, *convs):
        """[C] device vector for check_channels: per channel of a tensor the weight mass of the convs that read it -- sum |w| over the taps, max
        over the output channels -- normalised per conv to its largest INPUT channel (all of them, not only the slice), max over the convs.
        `convs`: OIHW weight tensors, or (w, lo, hi) when the tensor is input channels lo..hi of that conv."""
        g = None
        for cv in convs:
            w, lo, hi = cv if isinstance(cv, (tuple, list)) else (cv, 0, cv.shape[1])
            m = w.detach().to("cpu").abs().double().sum(dim=(2, 3)).max(dim=0)[0]
            top = float(m.max())
            m = (m / top) if top > 0.0 else torch.ones_like(m)
            m = m[lo:hi]
            g = m if g is None else torch.maximum(g, m)
        return self.vec(g.float())

    def read_range_flag(self):
        """The flag word (and clear it).  One 4-byte device->host copy: synchronises the current stream."""
        v = int(self.range_flag.item())
        if v:
            self.range_flag.zero_()
        return v

    def fallback_ops(self):
        """A second HipOps on the same device whose fp32-accurate contraction mode is the exact three-term bf16 split (BFSR_SPLIT=bf16x3: fp32's
        exponent range, six products) -- what guard.run_guarded re-runs a pass on when the fp16-pair split met a value outside its range.
        Returns self when this object already runs that split (or the native fp32 MFMA)."""
        if self.conv_mode != "x3" or self.split != "f16x2":
            return self
        if self._fallback is None:
            fb = HipOps(self.device)
            fb.split, fb.conv_mode = "bf16x3", self.conv_mode
            fb.profile_keys, fb.profile, fb._keylog = self.profile_keys, self.profile, self._keylog
            self._fallback = fb
        return self._fallback

    def squeeze2d(self, x, y):
        xp, xbs, Cc, H, W = _view(x)
        yp, ybs, c2, h2, w2 = _view(y)
        assert (c2, h2, w2) == (4 * Cc, H // 2, W // 2)
        _lib.check(self._launch(("squeeze2d",) + tuple(x.shape), lambda: self.lib.bfsr_squeeze2d(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "squeeze2d")
        return y

    def unsqueeze2d(self, x, y):
        xp, xbs, Cc, H, W = _view(x)
        yp, ybs, c2, h2, w2 = _view(y)
        assert (c2 * 4, h2, w2) == (Cc, 2 * H, 2 * W)
        _lib.check(self._launch(("unsqueeze2d",) + tuple(x.shape), lambda: self.lib.bfsr_unsqueeze2d(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "unsqueeze2d")
        return y

    def split2d(self, h, src, dst, reverse):
        hp, hbs, c2, H, W = _view(h)
        sp, sbs, Cc, _, _ = _view(src)
        dp, dbs, _, _, _ = _view(dst)
        assert c2 == 2 * Cc and dst.shape == src.shape
        _lib.check(self._launch(("split2d",) + tuple(src.shape), lambda: self.lib.bfsr_split2d(hp, hbs, sp, sbs, dp, dbs, src.shape[0], Cc, H, W, int(bool(reverse)),
                                         self._stream())), "split2d")
        return dst

    def standardize(self, x, y):
        xp, xbs, Cc, H, W = _view(x)
        yp, ybs, _, _, _ = _view(y)
        _lib.check(self._launch(("standardize",) + tuple(x.shape), lambda: self.lib.bfsr_standardize(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "standardize")
        return y

    def resize(self, x, y, mode, r_h, r_w, window=None):
        """window = (oy0, ox0, RH, RW): the resized image occupies that sub-window of y, rest zero."""
        xp, xbs, Cc, IH, IW = _view(x)
        yp, ybs, c2, OH, OW = _view(y)
        assert Cc == c2
        oy0, ox0, RH, RW = window if window is not None else (0, 0, OH, OW)
        _lib.check(self._launch(("resize", mode) + tuple(y.shape), lambda: self.lib.bfsr_resize(xp, xbs, IH, IW, yp, ybs, OH, OW, RH, RW, oy0, ox0, x.shape[0], Cc, mode,
                                        float(r_h), float(r_w), self._stream())), "resize")
        return y

    def resize_h2(self, x, out, mode, r_h, r_w, window=None):
        """resize(x) written as an h2 tensor (= resize + h2_pack, the same bits): x fp32 [B,C,IH,IW] view, out an h2 view [B,C/8,2,OH,OW,8]."""
        xp, xbs, Cc, IH, IW = _view(x)
        yp, ybs, c2, OH, OW = self._h2view(out, "resize_h2.out")
        assert Cc == c2 and x.shape[0] == out.shape[0]
        oy0, ox0, RH, RW = window if window is not None else (0, 0, OH, OW)
        _lib.check(self._launch(("resize_h2", mode, x.shape[0], Cc, OH, OW), lambda: self.lib.bfsr_resize_h2(
            xp, xbs, IH, IW, yp, ybs, OH, OW, RH, RW, oy0, ox0, x.shape[0], Cc, mode, float(r_h), float(r_w), self.range_flag.data_ptr(), self._stream())), "resize_h2")
        return out

    def maxpool2_h2(self, x, out_h2=None, out_f32=None):
        """2 x 2 max-pool of an h2 view into an h2 view and / or an fp32 tensor (= h2_unpack + maxpool2 [+ h2_pack], the same bits)."""
        xp, xbs, Cc, H, W = self._h2view(x, "maxpool2_h2.x")
        hp, hbs = (None, 0)
        if out_h2 is not None:
            hp, hbs, c2, h2, w2 = self._h2view(out_h2, "maxpool2_h2.out")
            assert (c2, h2, w2) == (Cc, H // 2, W // 2)
        fp, fbs = (None, 0)
        if out_f32 is not None:
            fp, fbs, c3, h3, w3 = _view(out_f32, "maxpool2_h2.out_f32")
            assert (c3, h3, w3) == (Cc, H // 2, W // 2)
        _lib.check(self._launch(("maxpool2_h2", x.shape[0], Cc, H, W), lambda: self.lib.bfsr_maxpool2_h2(
            xp, xbs, hp, hbs, fp, fbs, x.shape[0], Cc, H, W, self.range_flag.data_ptr(), self._stream())), "maxpool2_h2")
        return out_h2 if out_h2 is not None else out_f32

    def maxpool2(self, x, y):
        xp, xbs, Cc, H, W = _view(x)
        yp, ybs, _, _, _ = _view(y)
        _lib.check(self._launch(("maxpool2",) + tuple(x.shape), lambda: self.lib.bfsr_maxpool2(xp, xbs, yp, ybs, x.shape[0], Cc, H, W, self._stream())), "maxpool2")
        return y

    def axpb_clamp(self, x, y, a=1.0, b=0.0, lo=-3.4e38, hi=3.4e38, r=None):
        xp, xbs, Cc, H, W = _view(x)
        yp, ybs, _, _, _ = _view(y)
        rp, rbs = (None, 0)
        if r is not None:
            rp, rbs, _, _, _ = _view(r)
        _lib.check(self._launch(("axpb_clamp",) + tuple(x.shape), lambda: self.lib.bfsr_axpb_clamp(xp, xbs, rp, rbs, yp, ybs, x.shape[0], Cc, H, W, a, b, lo, hi,
                                            self._stream())), "axpb_clamp")
        return y

    # ---- LINF-LP ------------------------------------------------------------------------------------
    def linf_features(self, cf, coord, cell, phase, out, hidden):
        """cf [B,2*hidden,h,w], coord [B,qh,qw,2] contiguous, cell [B,2], phase [hidden/2,2] -> out [B,4*hidden,qh,qw]."""
        import numpy as np
        a = _lib.BfsrLinfFeatArgs()
        a.cf, a.cf_bs, c2, h, w = _view(cf, "linf.cf")
        a.out, a.out_bs, c4, qh, qw = _view(out, "linf.out")
        assert c2 == 2 * hidden and c4 == 4 * hidden and tuple(coord.shape) == (cf.shape[0], qh, qw, 2)
        assert coord.is_contiguous() and cell.is_contiguous() and phase.is_contiguous()
        a.coord, a.cell, a.phase = coord.data_ptr(), cell.data_ptr(), phase.data_ptr()
        a.B, a.hidden, a.h, a.w, a.qh, a.qw = cf.shape[0], hidden, h, w, qh, qw
        rx, ry, e = 2 / h / 2, 2 / w / 2, 1e-6            # linf.py:332-341 (python doubles -> float32 at the add)
        a.dy_neg, a.dy_pos, a.dx_neg, a.dx_pos = -1 * rx + e, 1 * rx + e, -1 * ry + e, 1 * ry + e
        a.clamp_lo, a.clamp_hi = -1 + 1e-6, 1 - 1e-6
        a.cy0, a.cy1, a.cx0, a.cx1 = -1 + 1.0 / h, 2 * (1.0 / h), -1 + 1.0 / w, 2 * (1.0 / w)
        _lib.check(self._launch(("linf_features",) + tuple(out.shape), lambda: self.lib.bfsr_linf_features(C.byref(a), self._stream())), "linf_features")
        return out

    def pack_linf_mlp(self, ws, bs, x3=True, quad_layers=None):
        """ws = [w1 [256,1024(,1,1)], w2, w3 [256,256], w4 [Cout,256]], bs = the four biases -> (packed weights, bias vector, Cout).
        quad_layers = (layers, D): the output is wanted in the private quad-major layout of linf_flow(ai_fmt=1) -- the last layer's
        rows are re-ordered so that every flow layer's D scales and D shifts each start at a multiple of four (zero rows as padding); the returned
        Cout is the padded row count and `linf_mlp` then writes [B][Cout/4][qh*qw][4]."""
        w = [t.detach().to("cpu", torch.float32).reshape(t.shape[0], t.shape[1]).contiguous() for t in ws]
        bs = [b.detach().to("cpu", torch.float32).reshape(-1) for b in bs]
        if quad_layers is not None:
            L, D = quad_layers
            S = (D + 3) // 4 * 4                    # per layer: S raw scales (D used), then S shifts (D used): both halves start on a quad
            blk = 2 * S
            assert w[3].shape[0] == 2 * D * L
            w4 = torch.zeros(L * blk, w[3].shape[1])
            b4 = torch.zeros(L * blk)
            for i in range(L):
                for h in (0, 1):
                    w4[i * blk + h * S: i * blk + h * S + D] = w[3][2 * D * i + h * D: 2 * D * i + (h + 1) * D]
                    b4[i * blk + h * S: i * blk + h * S + D] = bs[3][2 * D * i + h * D: 2 * D * i + (h + 1) * D]
            w[3], bs = w4.contiguous(), bs[:3] + [b4]
        hidden, Cout = w[1].shape[0], w[3].shape[0]
        mode = (2 if self.split == "f16x2" else 1) if x3 else 0         # the fp32-accurate mode follows BFSR_SPLIT
        n = self.lib.bfsr_linf_mlp_packed_size(hidden, Cout, mode)
        if n <= 0 or w[0].shape != (hidden, 4 * hidden) or w[2].shape != (hidden, hidden) or w[3].shape[1] != hidden:
            raise ValueError("pack_linf_mlp: unsupported MLP shape (hidden must be 256)")
        packed = torch.empty(n, dtype=torch.int16)
        scales = None
        if mode == 2:
            scales = [self.pow2_scale(t) for t in w]
            sc = (C.c_float * 4)(*scales)
            _lib.check(self.lib.bfsr_pack_linf_mlp_f16x2(w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(), hidden, Cout,
                                                         C.cast(sc, C.c_void_p), packed.data_ptr()), "pack_linf_mlp_f16x2")
        else:
            _lib.check(self.lib.bfsr_pack_linf_mlp(w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(), hidden, Cout, mode,
                                                   packed.data_ptr()), "pack_linf_mlp")
        bias = torch.cat(bs)
        fmt = 1 if quad_layers is not None else 0
        return packed.to(self.device), bias.to(self.device), Cout, (fmt if scales is None else (fmt, tuple(scales)))

    def linf_mlp(self, cf, coord, cell, phase, packed, out, hidden, x3=True, tile=None):
        """fused Fourier features + shared MLP: cf [B,2*hidden,h,w], coord [B,qh,qw,2], cell [B,2] -> out = affine_info [B,Cout,qh,qw]
        (the same buffer in the quad-major layout when the weights were packed with quad_layers)."""
        wts, bias, Cout, fmt = packed
        a = _lib.BfsrLinfMlpArgs()
        mode = 1 if x3 else 0
        if isinstance(fmt, tuple):                                       # two-term fp16 split: (layout, per-layer weight scales)
            fmt, scales = fmt
            mode = 2
            for i in range(4):
                a.acc_scale[i] = 1.0 / scales[i]
            a.flag = self.range_flag.data_ptr()
        a.out_fmt = fmt
        if cf.dtype == torch.float16:                                    # h2 tensor (hi + lo planes): 16-byte gathers
            a.cf, a.cf_bs, c2, h, w = self._h2view(cf, "linf_mlp.cf")
            a.cf_fmt = 1
        else:
            a.cf, a.cf_bs, c2, h, w = _view(cf, "linf_mlp.cf")
        a.out, a.out_bs, co, qh, qw = _view(out, "linf_mlp.out")
        assert c2 == 2 * hidden and co == Cout and tuple(coord.shape) == (cf.shape[0], qh, qw, 2)
        assert coord.is_contiguous() and cell.is_contiguous() and phase.is_contiguous()
        a.coord, a.cell, a.phase, a.wts, a.bias = coord.data_ptr(), cell.data_ptr(), phase.data_ptr(), wts.data_ptr(), bias.data_ptr()
        a.B, a.hidden, a.Cout, a.h, a.w, a.qh, a.qw = cf.shape[0], hidden, Cout, h, w, qh, qw
        rx, ry, e = 2 / h / 2, 2 / w / 2, 1e-6            # linf.py:332-341 (python doubles -> float32 at the add)
        a.dy_neg, a.dy_pos, a.dx_neg, a.dx_pos = -1 * rx + e, 1 * rx + e, -1 * ry + e, 1 * ry + e
        a.clamp_lo, a.clamp_hi = -1 + 1e-6, 1 - 1e-6
        a.cy0, a.cy1, a.cx0, a.cx1 = -1 + 1.0 / h, 2 * (1.0 / h), -1 + 1.0 / w, 2 * (1.0 / w)
        if fmt and (out.data_ptr() & 15 or a.out_bs & 3):
            raise ValueError("linf_mlp: the quad-major output needs a 16-byte aligned buffer")
        # 128-point workgroup tiles (round 6; fp16 and fp16-pair arithmetic: the activations of 128 points fit LDS): half the weight traffic per
        # point, identical bits -- and SLOWER (config 5: 19.1 against 15.6 ms, config 3: 8.6 against 8.2): at 181-220 registers one workgroup per CU
        # is left, and the kernel lives on the latency hiding of four waves per SIMD (profiles/r06m_*).  Off by default; BFSR_MLP_TILE=128 selects it.
        if tile is None:
            tile = int(os.environ.get("BFSR_MLP_TILE", "64"))
        a.tile = 128 if (tile == 128 and mode != 1 and qh * qw >= 128) else 64
        key = (("linf_mlp_f2" if mode == 2 else "linf_mlp_x3") if x3 else "linf_mlp_f16", hidden, Cout, cf.shape[0], qh, qw)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_linf_mlp(C.byref(a), mode, self._stream())), "linf_mlp")
        return out

    def logscale_sum(self, h, acc, coef=1.0, eps=1e-4):
        """acc[b] (float64 [B]) += coef * sum log(sigmoid(h[:, 1::2] + 2) + eps): an affine coupling's get_logdet(scale)."""
        hp, hbs, c2, H, W = _view(h, "logscale_sum.h")
        assert c2 % 2 == 0 and acc.dtype == torch.float64 and acc.numel() == h.shape[0] and acc.is_contiguous()
        _lib.check(self._launch(("logscale_sum", c2, h.shape[0], H, W),
                                lambda: self.lib.bfsr_logscale_sum(hp, hbs, h.shape[0], c2 // 2, H * W, eps, coef, acc.data_ptr(), self._stream())),
                   "logscale_sum")
        return acc

    def gaussian_logp(self, x, acc, h=None, coef=1.0):
        """acc[b] += coef * GaussianDiag.logp(mean, logs, x) with mean,logs = h[:, 0::2], h[:, 1::2] (standard normal if h is None)."""
        xp, xbs, Cx, H, W = _view(x, "gaussian_logp.x")
        hp, hbs = 0, 0
        if h is not None:
            hp, hbs, c2, hh, ww = _view(h, "gaussian_logp.h")
            assert (c2, hh, ww) == (2 * Cx, H, W)
        assert acc.dtype == torch.float64 and acc.numel() == x.shape[0] and acc.is_contiguous()
        _lib.check(self._launch(("gaussian_logp", Cx, x.shape[0], H, W),
                                lambda: self.lib.bfsr_gaussian_logp(xp, xbs, hp, hbs, x.shape[0], Cx, H * W, coef, acc.data_ptr(), self._stream())),
                   "gaussian_logp")
        return acc

    def zeros_f64(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    # ---- metrics / output formatting (metrics.hip) ----------------------------------------------------
    def resample_taps(self, x, y, idx, w, dim):
        """one separable-resampler pass along dim (0 rows / 1 cols) with tap tables idx int32 [O,P], w fp32 [O,P]."""
        xp, xbs, Cc, H, W = _view(x, "resample_taps.x")
        yp, ybs, _, OH, OW = _view(y, "resample_taps.y")
        O, P = idx.shape
        assert idx.dtype == torch.int32 and w.dtype == torch.float32 and w.shape == idx.shape and idx.is_contiguous() and w.is_contiguous()
        assert (OH, OW) == ((O, W) if dim == 0 else (H, O))
        _lib.check(self._launch(("resample_taps", dim) + tuple(x.shape) + (O, P), lambda: self.lib.bfsr_resample_taps(
            xp, xbs, yp, ybs, idx.data_ptr(), w.data_ptr(), x.shape[0], Cc, H, W, O, P, dim, self._stream())), "resample_taps")
        return y

    def sqdiff_sum(self, a, b, shave=0, luma=False, rgb_range=1.0):
        """float64 [B]: sum of ((a-b)/rgb_range)^2 over the window shaved by `shave` (luma: 3 channels -> Y first)."""
        ap, abs_, Cc, H, W = _view(a, "sqdiff_sum.a")
        bp, bbs, c2, h2, w2 = _view(b, "sqdiff_sum.b")
        assert (Cc, H, W) == (c2, h2, w2) and a.shape[0] == b.shape[0]
        out = self.zeros_f64(a.shape[0])
        _lib.check(self._launch(("sqdiff_sum",) + tuple(a.shape), lambda: self.lib.bfsr_sqdiff_sum(
            ap, abs_, bp, bbs, a.shape[0], Cc, H, W, int(shave), int(bool(luma)), float(rgb_range), out.data_ptr(), self._stream())), "sqdiff_sum")
        return out

    def ssim_sum(self, a, b, window121, scale=255.0):
        """float64 [B,C]: sum of the SSIM map over the valid region (11x11 window given as float64 [121] on the device)."""
        ap, abs_, Cc, H, W = _view(a, "ssim_sum.a")
        bp, bbs, c2, h2, w2 = _view(b, "ssim_sum.b")
        assert (Cc, H, W) == (c2, h2, w2) and window121.dtype == torch.float64 and window121.numel() == 121
        out = self.zeros_f64(a.shape[0] * Cc)
        _lib.check(self._launch(("ssim_sum",) + tuple(a.shape), lambda: self.lib.bfsr_ssim_sum(
            ap, abs_, bp, bbs, a.shape[0], Cc, H, W, float(scale), window121.data_ptr(), out.data_ptr(), self._stream())), "ssim_sum")
        return out.view(a.shape[0], Cc)

    def ssim_sum_w(self, a, b, window, cov_norm=1.0, scale=1.0):
        """float64 [B,C]: sum of the SSIM map over the valid region for a ws x ws window (float64 [ws*ws] on the device, ws <= 11), variances and
        covariance multiplied by cov_norm (metrics.hip, bfsr_ssim_sum_w)."""
        ap, abs_, Cc, H, W = _view(a, "ssim_sum_w.a")
        bp, bbs, c2, h2, w2 = _view(b, "ssim_sum_w.b")
        ws = int(round(window.numel() ** 0.5))
        assert (Cc, H, W) == (c2, h2, w2) and window.dtype == torch.float64 and ws * ws == window.numel()
        out = self.zeros_f64(a.shape[0] * Cc)
        _lib.check(self._launch(("ssim_sum_w", ws) + tuple(a.shape), lambda: self.lib.bfsr_ssim_sum_w(
            ap, abs_, bp, bbs, a.shape[0], Cc, H, W, float(scale), ws, window.data_ptr(), float(cov_norm), out.data_ptr(), self._stream())), "ssim_sum_w")
        return out.view(a.shape[0], Cc)

    def to_uint8(self, x):
        """uint8 [B,C,H,W] = round(clamp(x,0,1)*255), half-to-even."""
        xp, xbs, Cc, H, W = _view(x, "to_uint8.x")
        y = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        _lib.check(self._launch(("to_uint8",) + tuple(x.shape), lambda: self.lib.bfsr_to_uint8(
            xp, xbs, y.data_ptr(), x.shape[0], Cc * H * W, self._stream())), "to_uint8")
        return y

    def linf_flow(self, x, ai, y, lin_w, lin_b, layers, reverse, eps=1e-4, log_p=None, logdet_const=0.0, ai_fmt=0):
        a = _lib.BfsrLinfFlowArgs()
        a.ai_fmt = int(ai_fmt)
        if log_p is not None:
            assert not reverse and log_p.is_contiguous() and log_p.numel() == x.shape[0] * x.shape[2] * x.shape[3]
            a.log_p, a.logdet_const = log_p.data_ptr(), float(logdet_const)
        a.x, a.x_bs, D, qh, qw = _view(x, "linf_flow.x")
        a.ai, a.ai_bs, ca, _, _ = _view(ai, "linf_flow.ai")
        a.y, a.y_bs, _, _, _ = _view(y, "linf_flow.y")
        assert ca == ((2 * ((D + 3) // 4 * 4)) * layers if ai_fmt else 2 * D * layers)
        assert lin_w.numel() == (layers + 1) * D * D and lin_b.numel() == (layers + 1) * D
        a.lin_w, a.lin_b = lin_w.data_ptr(), lin_b.data_ptr()
        mode = int(reverse)                      # 0 forward, 1 (True) inverse, 2 = VJP of the inverse w.r.t. its input
        assert mode in (0, 1, 2)
        a.B, a.D, a.layers, a.qh, a.qw, a.reverse, a.eps = x.shape[0], D, layers, qh, qw, mode, eps
        key = ("linf_flow", mode, D, x.shape[0], qh, qw)
        _lib.check(self._launch(key, lambda: self.lib.bfsr_linf_flow(C.byref(a), self._stream())), "linf_flow(D=%d)" % D)
        return y

    def patch_fold(self, p, img, ps):
        pp, pbs, cp, qh, qw = _view(p)
        ip, ibs, Cc, H, W = _view(img)
        assert cp == Cc * ps * ps
        _lib.check(self._launch(("patch_fold",) + tuple(img.shape), lambda: self.lib.bfsr_patch_fold(pp, pbs, ip, ibs, p.shape[0], Cc, qh, qw, H, W, ps, self._stream())), "patch_fold")
        return img

    def linf_fold_skip(self, p, inp, H, W, ps, raw=None, out=None):
        """The tail of the LINF-LP harness in one launch (LINF-LP/test.py:168-171, 217): raw = fold(p)[.., :H, :W] + bilinear(inp -> H x W),
        out = clamp(0.5 raw + 0.5, 0, 1); p [B,C*ps*ps,qh,qw], inp [B,C,h,w] normalised; raw / out: [B,C,H,W] tensors or None (at least one).
        Bit-identical to patch_fold + resize + two axpb_clamp launches."""
        pp, pbs, cp, qh, qw = _view(p)
        ip, ibs, Cc, h, w = _view(inp)
        assert cp == Cc * ps * ps and (raw is not None or out is not None)
        rp, rbs = (None, 0) if raw is None else _view(raw)[:2]
        op_, obs = (None, 0) if out is None else _view(out)[:2]
        for t in (raw, out):
            assert t is None or tuple(t.shape) == (p.shape[0], Cc, H, W)
        _lib.check(self._launch(("linf_fold_skip", p.shape[0], Cc, H, W), lambda: self.lib.bfsr_linf_fold_skip(
            pp, pbs, ip, ibs, rp, rbs, op_, obs, p.shape[0], Cc, qh, qw, H, W, ps, h, w, float(h) / H, float(w) / W, self._stream())), "linf_fold_skip")
        return raw, out

    def linf_prep_residual(self, inp01, hr_hw, ps, qh, qw):
        """The LR-upsample residual of the dataset wrappers, unfolded, from the LR image alone (datasets/wrappers.py:203-228):
        gt [B,C*ps*ps,qh,qw] = unfold(zero-pad(lr_up - up(down(lr_up)))), lr_up = bilinear(2 inp01 - 1 -> H x W).  Two launches (the LR-sized `down`, then gt);
        bit-identical to axpb_clamp + three resize + axpb_clamp + patch_unfold."""
        H, W = hr_hw
        ip, ibs, Cc, h, w = _view(inp01)
        B = inp01.shape[0]
        down = self.empty(B, Cc, h, w)
        gt = self.empty(B, Cc * ps * ps, qh, qw)
        dp, dbs = _view(down)[:2]
        gp, gbs = _view(gt)[:2]
        ru_h, ru_w, rd_h, rd_w = float(h) / H, float(w) / W, float(H) / h, float(W) / w
        _lib.check(self._launch(("linf_prep_down", B, Cc, h, w), lambda: self.lib.bfsr_linf_prep_down(ip, ibs, dp, dbs, B, Cc, h, w, H, W, ru_h, ru_w, rd_h, rd_w,
                                                                                                       self._stream())), "linf_prep_down")
        _lib.check(self._launch(("linf_prep_residual", B, Cc, H, W), lambda: self.lib.bfsr_linf_prep_residual(ip, ibs, dp, dbs, gp, gbs, B, Cc, h, w, H, W, qh, qw, ps,
                                                                                                               ru_h, ru_w, self._stream())), "linf_prep_residual")
        return gt

    def patch_unfold(self, img, p, ps):
        pp, pbs, cp, qh, qw = _view(p)
        ip, ibs, Cc, H, W = _view(img)
        assert cp == Cc * ps * ps
        _lib.check(self._launch(("patch_unfold",) + tuple(img.shape), lambda: self.lib.bfsr_patch_unfold(ip, ibs, pp, pbs, p.shape[0], Cc, qh, qw, H, W, ps, self._stream())), "patch_unfold")
        return p

    def conv_direct(self, x, w, bias, y, stride, pad, act=ACT_NONE, slope=0.2):
        """w [Cout,Cin,KS,KS] contiguous device tensor."""
        xp, xbs, Cin, H, W = _view(x)
        yp, ybs, Cout, OH, OW = _view(y)
        KS = w.shape[2]
        assert w.is_contiguous() and tuple(w.shape[:2]) == (Cout, Cin)
        assert OH == (H + 2 * pad - KS) // stride + 1 and OW == (W + 2 * pad - KS) // stride + 1
        _lib.check(self._launch(("conv_direct",) + tuple(y.shape), lambda: self.lib.bfsr_conv2d_direct(xp, xbs, w.data_ptr(), _ptr(bias), yp, ybs, x.shape[0], Cin, Cout, H, W, KS,
                                               stride, pad, act, slope, self._stream())), "conv2d_direct")
        return y

    def grid_sample_add(self, x, coord, acc, out):
        """out = acc + grid_sample(x, coord(y,x), bilinear, border, align_corners=False); acc/out [B,C,qh,qw]."""
        xp, xbs, Cc, h, w = _view(x)
        ap, abs_, c2, qh, qw = _view(acc)
        op, obs, _, _, _ = _view(out)
        assert c2 == Cc and tuple(coord.shape) == (x.shape[0], qh, qw, 2) and coord.is_contiguous()
        _lib.check(self._launch(("grid_sample_add",) + tuple(out.shape), lambda: self.lib.bfsr_grid_sample_add(
            xp, xbs, coord.data_ptr(), ap, abs_, op, obs, x.shape[0], Cc, h, w, qh, qw, self._stream())), "grid_sample_add")
        return out
