"""Host-side schedule of SRFlow-LP on the HIP kernels (result-preserving restructuring of the
reference):

  * RRDB conditioning features are computed ONCE per LR batch (the reference recomputes the whole
    RRDB in decode, SRFlowNet_arch.py:152-153) and the heads no flow level reads (`upconv2`/`HRconv`/
    `conv_last` for 4x) are never computed;
  * everything in the coupling nets that depends only on the conditioning features `ft` -- the whole
    fFeatures net and the `ft` rows of fAffine's first conv (96 % of the flow MACs at level 1,
    FlowAffineCouplingsAblation.py:108-119) -- is hoisted out of the sequential FlowStep chain,
    batched over the K steps of a level and shared by encode and decode;
  * inverse 1x1-conv weights are inverted once (fp64 -> fp32, Permutations.py:41 does it per call);
  * growing concats (RDB, DenseBlock, `cat[z1, ft]`, `cat[skip, up]`) are never materialised: producers
    write into channel slices of one buffer, consumers read channel-slice views.

`ops` is the kernel backend (bfsr_amd.ops.HipOps).  No arithmetic happens in this file.
"""
import contextlib
import os

import torch

from . import spec
from .. import rng
from .options import opt_get
from ..ops import ACT_LRELU, ACT_NONE, ACT_RELU, MODE_BILINEAR, MODE_BILINEAR_AC, MODE_NEAREST

FUSED_COUPLING_C = (12, 24)                                     # flow widths the coupling_head / coupling_tail pair is built for
# BFSR_COUPLING: "fused" (default) = coupling_head / coupling_tail pair with the quad-major hand-over of pre_aff / h_ft; "fused-nchw" = the
# pair on NCHW tensors; "unfused" = generic launches for the sequential part (A/B and parity reference)
_COUPLING_MODE = os.environ.get("BFSR_COUPLING", "fused")
if _COUPLING_MODE not in ("fused", "fused-nchw", "unfused"):
    raise ValueError("BFSR_COUPLING must be 'fused', 'fused-nchw' or 'unfused'")
_QUADS = _COUPLING_MODE == "fused"
# fea_up{k} lives at LR resolution * 2^shift
_KEY_SHIFT = {"fea_up0": -1, "fea_up1": 0, "fea_up2": 1, "fea_up4": 2, "fea_up8": 3}


class _ConvP(object):
    """A packed conv + its per-channel epilogue vectors (all device tensors)."""

    def __init__(self, ops, w, bias=None, aff_shift=None, aff_scale=None, aff_post=None, post_scale=None, mtile=None,
                 f16=False, x3=None):
        """Contraction mode: 'f32' = native fp32 MFMA; 'x3' = fp32-accurate split on the 16-bit MFMA (ops.split: fp16 pair or bf16 triple; the default
        for 3x3 convs with >= 32 input channels, where it is 1.4-1.7x faster; x3=False pins fp32, e.g. for the fused
        two-stage kernel); 'f16' = reduced precision (LINF precision='fp16' only).  Epilogue and tensors are fp32."""
        pinned_f32 = x3 is False
        if x3 is None:
            x3 = getattr(ops, "conv_mode", "f32") == "x3" and w.shape[2] == 3 and w.shape[1] >= 32
        self.mode = "f16" if f16 else ("x3" if x3 else "f32")
        # wide 1x1 convs (the LINF MLP): GEMM kernel with 256 output channels per workgroup, in the fp16 or the x3 arithmetic
        # (x3 mode only for the fp16-precision models for now: the x3 arithmetic of this kernel is not faster than the fp32 MFMA)
        self.wide1x1 = (w.shape[2] == 1 and w.shape[0] >= 128 and w.shape[1] >= 64 and not pinned_f32 and f16)
        if self.wide1x1:
            self.pw = ops.pack_conv1x1(w, x3=not f16)
        else:
            self.pw = {"f16": ops.pack_conv_f16, "x3": ops.pack_conv_x3, "f32": ops.pack_conv}[self.mode](w, mtile)
        self.epi = ops.pack_epilogue(self.pw.Cout, bias, aff_shift, aff_scale, aff_post, post_scale)

    def run(self, ops, x, out, **kw):
        if self.wide1x1:
            return ops.conv1x1(x, self.pw, out, x3=self.mode != "f16", epi=self.epi, **kw)
        if self.mode == "f16":
            return ops.conv_f16(x, self.pw, out, epi=self.epi, **kw)
        if self.mode == "x3":
            return ops.conv_x3(x, self.pw, out, epi=self.epi, **kw)
        return ops.conv(x, self.pw, out, epi=self.epi, **kw)

    def run_h2(self, ops, x_h2, out, **kw):
        """The same conv over an h2 view of its input on the LDS-DMA kernels: conv_h2x for the split contraction, conv_h2s (reads the hi plane =
        the fp16 rounding of the activation, what conv_f16 stages) for precision='fp16'; `out`: an h2 view or an fp32 NCHW view.  `lo=True` keeps the
        lo plane of an fp16-mode h2 output (for an output that is unpacked to fp32 again)."""
        lo = kw.pop("lo", False)
        if self.mode == "f16":
            if getattr(self, "_h2s", None) is None:
                self._h2s = ops.pack_conv_h2s(self.pw._w)
            return ops.conv_h2s(x_h2, self._h2s, out, epi=self.epi, hi_only=(out.dtype == torch.float16 and not lo), **kw)
        if self.mode != "x3":
            raise ValueError("run_h2: no h2 kernel for contraction mode %r" % self.mode)
        return ops.conv_h2x(x_h2, self.pw, out, epi=self.epi, **kw)


def h2_mode(ops, f16):
    """Which LDS-DMA conv family a module's 3x3 convs can use over h2 tensors: 'h2s' (precision fp16), 'h2x' (fp16-pair split) or None."""
    if f16:
        return "h2s" if hasattr(ops, "conv_h2s") else None
    if getattr(ops, "conv_mode", "f32") == "x3" and getattr(ops, "split", "") == "f16x2" and hasattr(ops, "conv_h2x"):
        return "h2x"
    return None


def _chk(ops, t, gain=None, **kw):
    """Per-sample, per-channel dynamic-range check of an fp32 tensor that enters an fp16-pair region (ops.check_channels; a no-op on other splits /
    backends).  gain: _gain(ops, ...) of the convs that read t."""
    f = getattr(ops, "check_channels", None)
    if f is not None:
        f(t, gain, **kw) if (gain is not None or kw) else f(t)
    return t


def _gain(ops, *convs):
    """ops.channel_gain(...) where the backend has it (HipOps), else None (the check then treats every channel alike)."""
    f = getattr(ops, "channel_gain", None)
    return f(*convs) if f is not None and convs else None


class _Workspace(object):
    """Named scratch buffers, reallocated only when the requested shape changes."""

    def __init__(self, ops):
        self.ops, self.bufs = ops, {}

    def get(self, name, *shape):
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.ops.empty(*shape)
            self.bufs[name] = t
        return t


class RRDBEncoder(object):
    """RRDBNet trunk (RRDBNet_arch.py:67-148 / LINF-LP/models/rrdb.py:77-116): conv_first, nb x RRDB
    (3 x RDB of five 3x3 convs), trunk_conv + skip.  `taps` = RRDB indices whose output is wanted.

    Default path (split contraction): the dense blocks live in HBM as split tensors (ops.x3_empty: h2 = fp16 hi + lo under
    BFSR_SPLIT=f16x2, x3 = exact 3-term bf16 under bf16x3) and every RDB conv runs on the LDS-DMA kernel of that split (conv_h2x /
    conv_x3s: staging by dedicated loader waves); conv_first's output is packed
    once, tapped block outputs and the trunk output are unpacked / written as fp32.  `BFSR_CONV=f32` (or a backend without
    conv_x3s, or precision='fp16') keeps fp32 NCHW block buffers and the register-staged kernels."""

    def __init__(self, ops, sd, prefix, nb, nf=64, gc=32, skip_from_first=False, f16=False):
        # skip_from_first: LINF's RRDBNet adds the conv_first output (`fea = fea + trunk`, LINF-LP/models/rrdb.py:105-107)
        # whereas SRFlow's adds the trunk output (`last_lr_fea = fea + trunk` after the loop rebinds `fea`,
        # RRDBNet_arch.py:92-103)
        self.ops, self.nb, self.nf, self.gc, self.skip_from_first = ops, nb, nf, gc, skip_from_first
        packed_ok = nf % 16 == 0 and gc % 16 == 0
        self.x3s = not f16 and getattr(ops, "conv_mode", "f32") == "x3" and hasattr(ops, "conv_x3s") and packed_ok
        self.h2s = f16 and hasattr(ops, "conv_h2s") and packed_ok and nf % 32 == 0 and gc % 32 == 0
        g = lambda n: sd[prefix + n]
        mk = ((lambda w, b: _ConvX3S(ops, w, b)) if self.x3s else (lambda w, b: _ConvH2S(ops, w, b)) if self.h2s
              else (lambda w, b: _ConvP(ops, w, b, f16=f16)))
        self.conv_first = _ConvP(ops, g("conv_first.weight"), g("conv_first.bias"), f16=f16)
        self.blocks = []
        for b in range(nb):
            rdbs = []
            for r in (1, 2, 3):
                p = "RRDB_trunk.%d.RDB%d." % (b, r)
                rdbs.append([mk(g(p + "conv%d.weight" % i), g(p + "conv%d.bias" % i)) for i in range(1, 6)])
            self.blocks.append(rdbs)
        self.trunk_conv = mk(g("trunk_conv.weight"), g("trunk_conv.bias"))
        self.ws = _Workspace(ops)
        # range check of the two fp32 tensors at the ends of the packed region (ops.check_channels): conv_first's output is read by the five convs of
        # the first dense block (and, as the LINF skip, by nothing else that contracts it); `out_gain` = the convs that read the trunk output, set by
        # the owner (SRFlowEngine: upconv1 + the key rows of the hoisted convs; LINF: coef | freq)
        self.first_gain = _gain(ops, *[(g("RRDB_trunk.0.RDB1.conv%d.weight" % i), 0, nf) for i in range(1, 6)]) if nb > 0 else None
        self.out_gain = None

    def forward(self, x, out, on_block=None, taps=None):
        """x [B,3,h,w] -> out (a [B,nf,h,w] view) = fea + trunk_conv(fea).  `on_block(idx, fea_view, b0, b1)` is called after RRDB idx (for the samples b0..b1 the view holds)
        (only for idx in `taps` when given) with an fp32 view that is only valid during the call."""
        if self.x3s and self._use_chain(x):
            return self._forward_chain(x, out, on_block, taps)
        if self.x3s or self.h2s:
            return self._forward_packed(x, out, on_block, taps)
        ops, nf, gc = self.ops, self.nf, self.gc
        B, _, h, w = x.shape
        ring = [self.ws.get("dense%d" % i, B, nf + 4 * gc, h, w) for i in range(4)]
        cur = 0
        self.conv_first.run(ops, x, ring[cur][:, :nf])
        first = None
        if self.skip_from_first:
            first = self.ws.get("first", B, nf, h, w)
            ops.axpb_clamp(ring[cur][:, :nf], first)
        for idx, rdbs in enumerate(self.blocks):
            x_rrdb = ring[cur][:, :nf]
            for r, convs in enumerate(rdbs):
                D = ring[cur]
                for i in range(4):          # conv1..4: bias + LeakyReLU, written into the next 32-ch slice
                    convs[i].run(ops, D[:, :nf + i * gc], D[:, nf + i * gc: nf + (i + 1) * gc], act=ACT_LRELU, slope=0.2)
                nxt = (cur + 1) % 4
                if r < 2:                   # x5*0.2 + x
                    convs[4].run(ops, D, ring[nxt][:, :nf], res1=D[:, :nf], alpha1=0.2)
                else:                       # (x5*0.2 + x)*0.2 + x_rrdb
                    convs[4].run(ops, D, ring[nxt][:, :nf], res1=D[:, :nf], alpha1=0.2, res2=x_rrdb, alpha2=0.2)
                cur = nxt
            if on_block is not None and (taps is None or idx in taps):
                on_block(idx, ring[cur][:, :nf], 0, B)
        fea = ring[cur][:, :nf]
        self.trunk_conv.run(ops, fea, out, res1=first if first is not None else fea, alpha1=1.0)   # skip + trunk
        return out

    def _use_chain(self, x):
        """The fused chain (ops.conv_chain, conv_chain.hip: every dense-block conv + trunk_conv in ONE persistent launch) computes the bits of the
        per-launch kernel conv_h2x, so choosing between them is a pure scheduling decision.  It needs the fp16-pair split (h2 tensors) and at
        least one 16 x 32 tile per CU: the convs of a dense block are sequential per tile, so below that most workgroups would wait inside the
        launch and plain launch boundaries are cheaper (B = 1, 160^2: 99 us per dense block as launches, profiles/r05_d_chain_bench.txt).
        Measured under sustained load, interleaved A/B (profiles/r05_i_chain_ab.txt): 8 x 160^2 251 vs 286 us per dense block (two-stream
        launches), 64 x 96^2 742 vs 761, 16 x 256^2 1341 vs 1337.  BFSR_RRDB=launches keeps one launch per conv."""
        ops = self.ops
        if (os.environ.get("BFSR_RRDB", "chain") != "chain" or not hasattr(ops, "conv_chain") or getattr(ops, "split", None) != "f16x2"
                or getattr(ops, "chain_disabled", False)):      # (set by guard.run_guarded after a dependency time-out)
            return False
        B, _, h, w = x.shape
        cus, n_items = ops.cu_count(), B * ((h + 15) // 16) * ((w + 31) // 32)
        # ... and something to recover: the chip is power-limited under these kernels, so closing launch gaps alone buys nothing (16 x 256^2, 2048
        # tiles = 8.0 rounds: 1390 vs 1326 us per dense block over 69 blocks, profiles/r05_j_chain_ab_69.txt); the gain is the partly filled
        # last round of every conv (8 x 160^2: 400 tiles = 1.56 rounds)
        return n_items >= cus and -(-n_items // cus) * cus >= 1.1 * n_items

    def _forward_chain(self, x, out, on_block, taps):
        """RRDBNet_arch.py:89-103 / LINF-LP/models/rrdb.py:100-107 with all 15 * nb dense-block convs and trunk_conv as ONE launch of the
        chain kernel (conv_chain.hip): same buffers and views as _forward_packed (ring of four 192-channel h2 block buffers), tapped RRDB
        outputs leave the launch as a second fp32 copy of the tapped conv's result."""
        ops, nf, gc = self.ops, self.nf, self.gc
        B, _, h, w = x.shape
        o = lambda c: c // 8
        key = (B, h, w)
        if getattr(self, "_pkkey", None) != key:
            self._pkkey = key
            self._ring = [ops.h2_empty(B, nf + 4 * gc, h, w) for _ in range(4)]
            self._first = ops.h2_empty(B, nf, h, w) if self.skip_from_first else None
            self._chain = None
        ring = self._ring
        tmp = self.ws.get("x3_io", B, nf, h, w)
        want = [idx for idx in range(self.nb) if on_block is not None and (taps is None or idx in taps)]
        # the chain holds raw pointers: its last conv writes a buffer the encoder owns (`out` may be a fresh tensor on every call, e.g. LINF's
        # gen_feat) and one copy hands the result over
        trunk_out = self.ws.get("chain_out", B, nf, h, w)
        ckey = (trunk_out.data_ptr(), tuple(want))
        if getattr(self, "_chain", None) is None or self._chain[0] != ckey:
            tapbuf = {idx: self.ws.get("tap%d" % idx, B, nf, h, w) for idx in want}
            specs, cur = [], 0
            for idx, rdbs in enumerate(self.blocks):
                x_rrdb = ring[cur][:, :o(nf)]
                for r, convs in enumerate(rdbs):
                    D = ring[cur]
                    for i in range(4):
                        specs.append(dict(x=D[:, :o(nf + i * gc)], pw=convs[i].pw, out=D[:, o(nf + i * gc): o(nf + (i + 1) * gc)], epi=convs[i].epi,
                                          act=ACT_LRELU, slope=0.2))
                    nxt = (cur + 1) % 4
                    sp = dict(x=D, pw=convs[4].pw, out=ring[nxt][:, :o(nf)], epi=convs[4].epi, res1=D[:, :o(nf)], alpha1=0.2)
                    if r == 2:
                        sp.update(res2=x_rrdb, alpha2=0.2)
                        if idx in tapbuf:
                            sp["out2"] = tapbuf[idx]
                    specs.append(sp)
                    cur = nxt
            fea = ring[cur][:, :o(nf)]
            specs.append(dict(x=fea, pw=self.trunk_conv.pw, out=trunk_out, epi=self.trunk_conv.epi, res1=self._first if self.skip_from_first else fea, alpha1=1.0))
            self._chain = (ckey, [ops.conv_chain(specs)], tapbuf)
        _, chains, tapbuf = self._chain
        self.conv_first.run(ops, x, tmp)
        _chk(ops, tmp, self.first_gain)
        ops.h2_pack(tmp, ring[0][:, :o(nf)])
        if self.skip_from_first:
            ops.h2_pack(tmp, self._first)
        for ch in chains:
            ch.run()
        ops.axpb_clamp(trunk_out, out)
        for idx in want:
            on_block(idx, tapbuf[idx], 0, B)
        _chk(ops, out, self.out_gain)
        return out

    def _forward_packed(self, x, out, on_block, taps):
        """The dense blocks on packed 16-bit tensors: the split tensors of the fp32-accurate mode (h2 on conv_h2x or x3 on conv_x3s,
        see the class docstring) or, for precision='fp16', h2 (fp16
        hi + lo planes, conv_h2s: convs read hi, the residual chain reads hi + lo; x1..x4 are written hi-only)."""
        ops, nf, gc = self.ops, self.nf, self.gc
        B, _, h, w = x.shape
        o = lambda c: c // 8                                           # channel -> octet index of a packed tensor
        empty, pack, unpack = (ops.x3_empty, ops.x3_pack, ops.x3_unpack) if self.x3s else (ops.h2_empty, ops.h2_pack, ops.h2_unpack)
        inner = {"hi_only": True} if self.h2s else {}
        key = (B, h, w)
        if getattr(self, "_pkkey", None) != key:                       # packed block buffers (not fp32: kept outside _Workspace)
            self._pkkey = key
            self._ring = [empty(B, nf + 4 * gc, h, w) for _ in range(4)]
            self._first = empty(B, nf, h, w) if self.skip_from_first else None
        ring = self._ring
        tmp = self.ws.get("x3_io", B, nf, h, w)                        # fp32 staging at the two ends of the packed region

        def part(b0, b1):
            """the whole chain for samples [b0, b1) (every kernel is per-sample: a batch slice computes the same bits); a generator that
            yields after every launch so that two halves can be enqueued alternately"""
            rg = [r_[b0:b1] for r_ in ring]
            first = self._first[b0:b1] if self.skip_from_first else None
            xs, outs, tmps = x[b0:b1], out[b0:b1], tmp[b0:b1]
            cur = 0
            self.conv_first.run(ops, xs, tmps)
            if self.x3s:
                _chk(ops, tmps, self.first_gain)
            pack(tmps, rg[cur][:, :o(nf)])
            if self.skip_from_first:
                pack(tmps, first)
            yield
            for idx, rdbs in enumerate(self.blocks):
                x_rrdb = rg[cur][:, :o(nf)]
                for r, convs in enumerate(rdbs):
                    D = rg[cur]
                    for i in range(4):
                        convs[i].run(ops, D[:, :o(nf + i * gc)], D[:, o(nf + i * gc): o(nf + (i + 1) * gc)], act=ACT_LRELU, slope=0.2, **inner)
                        yield
                    nxt = (cur + 1) % 4
                    if r < 2:
                        convs[4].run(ops, D, rg[nxt][:, :o(nf)], res1=D[:, :o(nf)], alpha1=0.2)
                    else:
                        convs[4].run(ops, D, rg[nxt][:, :o(nf)], res1=D[:, :o(nf)], alpha1=0.2, res2=x_rrdb, alpha2=0.2)
                    cur = nxt
                    yield
                if on_block is not None and (taps is None or idx in taps):
                    on_block(idx, unpack(rg[cur][:, :o(nf)], tmps), b0, b1)
            fea = rg[cur][:, :o(nf)]
            self.trunk_conv.run(ops, fea, outs, res1=first if self.skip_from_first else fea, alpha1=1.0)
            if self.x3s:
                _chk(ops, outs, self.out_gain)
            yield

        # Tile quantisation: a dense-block conv of B x (h/16) x (w/32) tiles runs in ceil(tiles / CUs) rounds of one persistent workgroup per CU
        # (config 2: 400 tiles on 256 CUs = 2 rounds for 1.56 rounds of work, and a tile's cost does not shrink with its height).  When more than
        # ~20 % of the slots would idle, the batch is split in two halves whose chains run on two streams: the second half's workgroups take
        # the CUs the first half leaves free, and vice versa -- 324 -> 300 us per dense block at 8 x 160^2 (tools/exp/rdb_split.py; no gain and
        # therefore no split at config 4's 1152 tiles).  `side` is the caller's second stream (None: never split).
        side = getattr(self, "side_stream", None)
        n_items = B * ((h + 15) // 16) * ((w + 31) // 32)
        cus = ops.cu_count() if hasattr(ops, "cu_count") else 256       # one persistent workgroup per CU in the dense-block kernels
        rounds = -(-n_items // cus)
        if side is not None and B % 2 == 0 and n_items > cus and rounds * cus >= 1.2 * n_items:
            # the two halves are enqueued ALTERNATELY, launch by launch: with one half's whole chain enqueued before the other's the two streams
            # hardly overlapped (0.3 ms gained instead of 2.9 at config 2; the host is not the limit: ~10 us per launch)
            main = torch.cuda.current_stream(ops.device)
            side.wait_stream(main)
            ga, gb = part(0, B // 2), part(B // 2, B)
            live = True
            while live:
                live = next(ga, False) is not False
                with torch.cuda.stream(side):
                    live = (next(gb, False) is not False) or live
            main.wait_stream(side)
        else:
            for _ in part(0, B):
                pass
        return out


class _ConvX3S(object):
    """A 3x3 conv over split tensors (ops.conv_x3s -> conv_h2x or conv_x3s): weights packed for 32-cout workgroup tiles + bias epilogue."""

    def __init__(self, ops, w, bias=None):
        self.pw = ops.pack_conv_x3(w, 1, lazy=True)               # only the conv_h2x / conv_x3s packing is ever used
        self.epi = ops.pack_epilogue(self.pw.Cout, bias)

    def run(self, ops, x, out, **kw):
        return ops.conv_x3s(x, self.pw, out, epi=self.epi, **kw)


class _ConvH2S(object):
    """A 3x3 conv over h2 tensors (ops.conv_h2s): fp16 weights packed for 32-cout workgroup tiles + bias epilogue."""

    def __init__(self, ops, w, bias=None):
        self.pw = ops.pack_conv_h2s(w)
        self.epi = ops.pack_epilogue(self.pw.Cout, bias)

    def run(self, ops, x, out, **kw):
        return ops.conv_h2s(x, self.pw, out, epi=self.epi, **kw)


class _CouplingStep(object):
    """Device-side parameters of one FlowStep (FlowStep.py:31-86)."""


class SRFlowEngine(object):
    def __init__(self, opt, sd, ops, nb=None):
        self.opt, self.ops = opt, ops
        g = opt["network_G"]
        self.scale = opt["scale"]
        self.nb = nb if nb is not None else g["nb"]
        self.layers = spec.flow_layers(opt)
        self.level_names = spec.level_to_name(self.scale)
        self.L = g["flow"]["L"]
        self.block_idxs = list(opt_get(opt, ["network_G", "flow", "stackRRDB", "blocks"]) or [])
        self.concat = bool(opt_get(opt, ["network_G", "flow", "stackRRDB", "concat"]))
        self.n_cond = spec.n_rrdb_channels(opt) if self.concat else 64
        if self.n_cond != spec.IN_CHANNELS_RRDB:
            raise NotImplementedError("coupling nets hard-code 320 conditioning channels "
                                      "(FlowAffineCouplingsAblation.py:30); got %d" % self.n_cond)
        self.ws = _Workspace(ops)
        self._cond_key, self._cond = None, None
        self._side_stream = None
        self._hid = {}                  # h2 tensors between coupling_head and coupling_tail, per (direction, level)
        self._wide_w = {}               # data_ptr of a step's W / W^-1 -> its K-permuted copy for coupling_wide_tail
        self._z1h_next = {}             # tag -> (z.data_ptr(), layer index): the h2 copy of z1 the last wide tail wrote is valid for exactly that step
        self._load(sd)

    # ------------------------------------------------------------------------------------------
    def _load(self, sd):
        ops = self.ops
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        self.rrdb = RRDBEncoder(ops, sd, "RRDB.", self.nb, nf=self.opt["network_G"]["nf"])
        need = set(self.level_names[l] for l in range(1, self.L + 1))
        self.need_keys = need
        g = lambda n: sd["RRDB." + n]
        self.upconvs = {}
        chain = [("fea_up2", "upconv1"), ("fea_up4", "upconv2"), ("fea_up8", "upconv3")]
        deepest = max([i for i, (k, _) in enumerate(chain) if k in need], default=-1)
        for i in range(deepest + 1):
            k, n = chain[i]
            self.upconvs[k] = _ConvP(ops, g(n + ".weight"), g(n + ".bias"))
        if "fea_up0" in need and not opt_get(self.opt, ["network_G", "flow", "fea_up0"]):
            raise ValueError("flow level needs fea_up0 but network_G.flow.fea_up0 is not enabled")

        P = "flowUpsamplerNet.layers.%d."
        self.steps = {}
        self.splits = {}
        for ly in self.layers:
            p = P % ly.index
            if ly.type == "step":
                st = _CouplingStep()
                C = ly.C
                W = sd[p + "invconv.weight"]
                Winv = torch.inverse(W.double()).float()                     # Permutations.py:41, once
                st.w_fwd, st.w_inv = ops.vec(W), ops.vec(Winv)
                st.w_fwd_t, st.w_inv_t = ops.vec(W.t().contiguous()), ops.vec(Winv.t().contiguous())
                logs = sd[p + "actnorm.logs"].reshape(-1)
                st.an_bias = ops.vec(sd[p + "actnorm.bias"])
                st.an_exp = ops.vec(torch.exp(logs))
                st.an_expneg = ops.vec(torch.exp(-logs))
                # per-pixel logdet of actnorm + invconv (FlowActNorms.py:85-91, Permutations.py:37)
                st.ld_const = float(logs.double().sum()) + float(torch.slogdet(W.detach().cpu().double())[1])
                if C == 96 and hasattr(ops, "pack_wide_wmat"):          # the K-permuted copies coupling_wide_tail contracts (coupled or not: a tail applies the NEXT step's W in encode)
                    self._wide_w[st.w_fwd.data_ptr()] = ops.pack_wide_wmat(W)
                    self._wide_w[st.w_inv.data_ptr()] = ops.pack_wide_wmat(Winv)
                if ly.coupled:
                    cn = C // 2
                    a = p + "affine.fAffine."
                    w0 = sd[a + "0.weight"]
                    st.aff0_z1 = _ConvP(ops, w0[:, :cn].contiguous(), x3=False, aff_shift=sd[a + "0.actnorm.bias"],
                                        aff_scale=torch.exp(sd[a + "0.actnorm.logs"]), mtile=2)
                    st.aff0_ft_w = w0[:, cn:].contiguous()                     # hoisted (batched per level)
                    st.aff2 = _ConvP(ops, sd[a + "2.weight"], aff_shift=sd[a + "2.actnorm.bias"],
                                     aff_scale=torch.exp(sd[a + "2.actnorm.logs"]), mtile=2)
                    st.aff4 = _ConvP(ops, sd[a + "4.weight"], bias=sd[a + "4.bias"],
                                     post_scale=torch.exp(sd[a + "4.logs"] * 3))
                    # levels with 12 / 24 flow channels: the whole sequential part of the step as TWO kernels: coupling_head (3x3 on z1 +
                    # hoisted partial, 1x1 chained in registers; coupling.hip) -> hid as an h2 tensor -> coupling_tail (Conv2dZeros on the LDS-DMA
                    # kernel with the pointwise chain as its epilogue; conv_h2s.hip).  Both run the two-term fp16 split, so the pair needs
                    # BFSR_SPLIT=f16x2 (the default); otherwise, and with BFSR_COUPLING=unfused, the generic launches are used
                    # (3x3 + fused 1x1, Conv2dZeros, flow_pointwise).  History of the pair: DESIGN.md section 5.
                    st.fused = (C in FUSED_COUPLING_C and w0.shape[0] == 64 and hasattr(ops, "coupling_head")
                                and getattr(ops, "conv_mode", "f32") == "x3" and getattr(ops, "split", "") == "f16x2"
                                and _COUPLING_MODE != "unfused")
                    if st.fused:
                        st.head = ops.pack_coupling_head(w0[:, :cn].contiguous(), sd[a + "2.weight"], sd[a + "0.actnorm.bias"],
                                                         torch.exp(sd[a + "0.actnorm.logs"]), sd[a + "2.actnorm.bias"],
                                                         torch.exp(sd[a + "2.actnorm.logs"]))
                        st.tail = ops.pack_coupling_tail(sd[a + "4.weight"], sd[a + "4.bias"], torch.exp(sd[a + "4.logs"] * 3))
                    # levels the pair does not cover (C = 96: the head's 3x3 weights for 48 z1 channels do not fit LDS beside a tile): the same
                    # three stages on the fp16-split kernels that exist -- split 3x3 on z1 with the hoisted partial as pre_add (raw result), the
                    # 1x1-only coupling_head (ActNorm + ReLU, 1x1, ActNorm + ReLU -> h2 tensor), Conv2dZeros on conv_h2x -- instead of the fused
                    # 3x3 -> 1x1 on the native fp32 MFMA + a register-staged Conv2dZeros: 96 -> 72 us per step at 8 x 80^2, 721 -> 575 us at 64 x 96^2
                    # (tools/exp/level3_bench.py)
                    st.chain = None
                    if (not st.fused and w0.shape[0] == 64 and cn % 16 == 0 and hasattr(ops, "coupling_head") and hasattr(ops, "conv_h2x")
                            and getattr(ops, "conv_mode", "f32") == "x3" and getattr(ops, "split", "") == "f16x2" and _COUPLING_MODE != "unfused"):
                        st.chain = (ops.pack_conv_x3(w0[:, :cn].contiguous(), 2),
                                    ops.pack_coupling_head(None, sd[a + "2.weight"], sd[a + "0.actnorm.bias"], torch.exp(sd[a + "0.actnorm.logs"]),
                                                           sd[a + "2.actnorm.bias"], torch.exp(sd[a + "2.actnorm.logs"])),
                                    ops.pack_conv_x3(sd[a + "4.weight"], 1, lazy=True), st.aff4.epi)
                    # Round 6: the same step as TWO streaming kernels (coupling_wide.hip): split 3x3 on the h2 copy of z1 + hoisted partial + 1x1 chained in
                    # registers -> hid (h2); Conv2dZeros 64 -> 96 with the whole pointwise chain as its epilogue, which also hands the next step its z1
                    # as an h2 tensor.  Five launches -> two per step (BFSR_WIDE=0 keeps the launches above; the log-det paths always do).
                    st.wide = None
                    if (st.chain is not None and C == 96 and hasattr(ops, "coupling_wide_head") and os.environ.get("BFSR_WIDE", "1") != "0"):
                        st.wide = ops.pack_coupling_wide(w0[:, :cn].contiguous(), sd[a + "2.weight"], sd[a + "0.actnorm.bias"], torch.exp(sd[a + "0.actnorm.logs"]),
                                                         sd[a + "2.actnorm.bias"], torch.exp(sd[a + "2.actnorm.logs"]), sd[a + "4.weight"], sd[a + "4.bias"],
                                                         torch.exp(sd[a + "4.logs"] * 3))
                    f = p + "affine.fFeatures."
                    st.ft0_w = sd[f + "0.weight"]
                    st.ft0_shift = sd[f + "0.actnorm.bias"].reshape(-1)
                    st.ft0_scale = torch.exp(sd[f + "0.actnorm.logs"]).reshape(-1)
                    st.ft2 = _ConvP(ops, sd[f + "2.weight"], aff_shift=sd[f + "2.actnorm.bias"],
                                    aff_scale=torch.exp(sd[f + "2.actnorm.logs"]))
                    st.ft4 = _ConvP(ops, sd[f + "4.weight"], bias=sd[f + "4.bias"],
                                    post_scale=torch.exp(sd[f + "4.logs"] * 3))
                    # fused levels: the rest of the hoisted fFeatures net on the coupling pair's kernels -- fFeatures.0's ActNorm + ReLU and
                    # fFeatures.2 on the 1x1-only form of coupling_head (h2 output), fFeatures.4 (Conv2dZeros 64 -> 2C) on the tail's conv
                    # kernel in groups of <= 32 output channels (bfsr_conv3x3_h2r)
                    st.fthead = st.ft4r = st.ft4x = None
                    if st.chain is not None:                                   # same kernels for the hoisted fFeatures net of these levels; Conv2dZeros 64 -> 2C in one conv_h2x launch
                        st.fthead = ops.pack_coupling_head(None, sd[f + "2.weight"], st.ft0_shift, st.ft0_scale, sd[f + "2.actnorm.bias"],
                                                           torch.exp(sd[f + "2.actnorm.logs"]))
                        st.ft4x = (ops.pack_conv_x3(sd[f + "4.weight"], 1, lazy=True), st.ft4.epi)
                    if st.fused and hasattr(ops, "conv_h2r") and (2 * C) % 24 == 0:
                        st.fthead = ops.pack_coupling_head(None, sd[f + "2.weight"], st.ft0_shift, st.ft0_scale, sd[f + "2.actnorm.bias"],
                                                           torch.exp(sd[f + "2.actnorm.logs"]))
                        w4, b4, p4 = sd[f + "4.weight"], sd[f + "4.bias"].reshape(-1), torch.exp(sd[f + "4.logs"] * 3).reshape(-1)
                        st.ft4r = [(ops.pack_coupling_tail(w4[g:g + 24].contiguous(), b4[g:g + 24], p4[g:g + 24]),
                                    ops.pack_epilogue(24, bias=b4[g:g + 24], post_scale=p4[g:g + 24]), g, g + 24) for g in range(0, 2 * C, 24)]
                self.steps[ly.index] = st
            elif ly.type == "split":
                self.splits[ly.index] = _ConvP(ops, sd[p + "conv.weight"], bias=sd[p + "conv.bias"],
                                               post_scale=torch.exp(sd[p + "conv.logs"] * 3))
        # batched hoisted first convs per level
        self.hoist = {}
        lvl_w = {}                                                      # level -> the two stacked first convs over the level's conditional (for the range-check gains below)
        for level in range(1, self.L + 1):
            idxs = [ly.index for ly in self.layers if ly.type == "step" and ly.coupled and ly.level == level]
            if not idxs:
                continue
            wf = torch.cat([self.steps[i].ft0_w for i in idxs], 0)
            sh = torch.cat([self.steps[i].ft0_shift for i in idxs], 0)
            sc = torch.cat([self.steps[i].ft0_scale for i in idxs], 0)
            wa = torch.cat([self.steps[i].aff0_ft_w for i in idxs], 0)
            lvl_w[level] = (wf, wa)
            hz = dict(idxs=idxs, up2=False)
            # quad-major hand-over (see _hoist_level): which hoisted tensors of this level only the coupling pair reads
            fused_all = all(getattr(self.steps[i], "fused", False) for i in idxs)
            pos_of = {ly.index: p_ for p_, ly in enumerate(self.layers)}
            def _prev_is_fused_step(i):
                pv = self.layers[pos_of[i] - 1] if pos_of[i] > 0 else None
                return (pv is not None and pv.type == "step" and pv.coupled and pv.level == level
                        and getattr(self.steps[pv.index], "fused", False))
            hz["hft_q4"] = set(i for i in idxs if getattr(self.steps[i], "fused", False) and _prev_is_fused_step(i))
            def _prev_is_wide_step(i):
                pv = self.layers[pos_of[i] - 1] if pos_of[i] > 0 else None
                return (pv is not None and pv.type == "step" and pv.coupled and pv.level == level and getattr(self.steps[pv.index], "wide", None) is not None)
            # ... and the same for the wide level's streaming pair (only while the level's pre_aff is an h2 tensor: see _hoist_level)
            hz["hft_q4_wide"] = set(i for i in idxs if getattr(self.steps[i], "wide", None) is not None and _prev_is_wide_step(i))
            # (the register-staged x4 taps kernel has no quad-major epilogue: without conv_up4_h2t the raw fFeatures.0 result of that level stays
            # NCHW and the 1x1-only head reads it so)
            h4t_ok = (self._taps_up2(level) == 2 and getattr(ops, "conv_mode", "f32") == "x3" and getattr(ops, "split", "") == "f16x2"
                      and hasattr(ops, "conv_up4_h2t") and bool(getattr(self.rrdb, "x3s", False)) and os.environ.get("BFSR_UP4", "h2t") == "h2t"
                      and wf.shape[0] % 32 == 0 and wa.shape[0] % 32 == 0 and (wf.shape[1] - 64) % 16 == 0)
            hz["ffast"] = all(self.steps[i].fthead is not None for i in idxs) and self._taps_up2(level) in (0, 1, 2)
            hz["pre_q4"] = (fused_all and (self._taps_up2(level) in (0, 1, False, None) or h4t_ok) and getattr(ops, "conv_mode", "f32") == "x3")
            # Round 3: the 64 -> 16*64 key convs of the finer levels run on conv_x3s (LDS-DMA staging by loader waves, persistent) over
            # an x3 copy of the key channels instead of the register-staged conv_bf16x3 kernel: 5.51 -> 4.80 ms at 8 x 320^2
            # (175 -> 201 TFLOP/s-equivalent).  The 320 -> 1024 hoists of the coarser levels were measured too and are NOT moved:
            # 5.17 -> 5.45 ms at 8 x 160^2 -- conv_x3s tiles 32 output channels per workgroup, so a 1024-channel conv re-stages every
            # input tile 32 times (33 GB through L2 per launch), conv_bf16x3's 64-channel tiles half as often.
            hz["x3s"] = bool(getattr(self.rrdb, "x3s", False))
            if self._taps_up2(level):
                # the 256 stacked-RRDB channels of this level are the LR-resolution taps upsampled x2: their share of the
                # 3x3 conv runs on the LR grid with parity pre-summed weights (4/9 of the MACs, nothing materialised);
                # the 64 native-resolution key channels are convolved into the same accumulators by the same kernel.
                hz.update(up2=True, up=self._taps_up2(level), x3=getattr(ops, "conv_mode", "f32") == "x3",
                          ft0_epi=ops.pack_epilogue(wf.shape[0], aff_shift=sh, aff_scale=sc))
                if hz["up"] == 2:       # x4: 25 pre-summed matrices (25 instead of 144 tap products per source pixel)
                    hz.update(ft0_taps=ops.pack_conv_up4_x3(wf[:, 64:].contiguous()), aff0_taps=ops.pack_conv_up4_x3(wa[:, 64:].contiguous()),
                              ft0_key=ops.pack_conv_x3(wf[:, :64].contiguous(), 2), aff0_key=ops.pack_conv_x3(wa[:, :64].contiguous(), 2))
                    # Round 5: the same 25 blocks on conv_up4_h2t (conv_up2_h2t's structure: taps split once into an h2 tensor, LDS-DMA staging,
                    # all nine phase classes per workgroup item, quad-major output -> the level gets the quad-major hand-over of the x2 levels)
                    if h4t_ok and hz["ffast"] and hz["pre_q4"]:
                        hz["h4t"] = (ops.pack_conv_up4_h2t(wf[:, 64:].contiguous()), ops.pack_conv_up4_h2t(wa[:, 64:].contiguous()))
                elif hz["x3"]:
                    # 3xBF16 kernels: key channels by the plain conv (no epilogue) into the output buffer, then the taps
                    # kernel adds them back through pre_add and applies the epilogue
                    hz.update(ft0_taps=ops.pack_conv_up2_x3(wf[:, 64:].contiguous()), aff0_taps=ops.pack_conv_up2_x3(wa[:, 64:].contiguous()),
                              ft0_key=ops.pack_conv_x3(wf[:, :64].contiguous(), 1 if hz["x3s"] else 2, lazy=True),
                              aff0_key=ops.pack_conv_x3(wa[:, :64].contiguous(), 1 if hz["x3s"] else 2, lazy=True))
                    # Round 4: with the quad-major hand-over the taps run on conv_up2_h2t (taps split once into an h2 tensor, LDS-DMA staging,
                    # all four output parities per workgroup item): 6.4 -> 4.8 ms per launch at 8 x 160^2 -> 320^2 (tools/exp/taps_bench.py)
                    if (hz["x3s"] and getattr(ops, "split", "") == "f16x2" and hasattr(ops, "conv_up2_h2t") and hz["ffast"] and hz["pre_q4"]
                            and wf.shape[0] % 32 == 0 and wa.shape[0] % 32 == 0 and (wf.shape[1] - 64) % 16 == 0):
                        hz["h2t"] = (ops.pack_conv_up2_h2t(wf[:, 64:].contiguous(), wf[:, :64].contiguous()),
                                     ops.pack_conv_up2_h2t(wa[:, 64:].contiguous(), wa[:, :64].contiguous()))
                else:
                    hz.update(ft0_taps=ops.pack_conv_up2(wf[:, 64:].contiguous()), aff0_taps=ops.pack_conv_up2(wa[:, 64:].contiguous()),
                              ft0_key=ops.pack_conv(wf[:, :64].contiguous(), 2), aff0_key=ops.pack_conv(wa[:, :64].contiguous(), 2))
            else:
                hz["x3s"] = False
                hz.update(ft0=_ConvP(ops, wf, aff_shift=sh, aff_scale=sc, mtile=2), aff0=_ConvP(ops, wa, mtile=2))
                if hz["ffast"]:
                    hz["ft0_raw"] = _ConvP(ops, wf, mtile=2)
            if hz["up2"] and not (hz.get("x3") and hz["up"] in (1, 2)):
                hz["x3s"] = False
            self.hoist[level] = hz
            for i in idxs:
                del self.steps[i].ft0_w, self.steps[i].aff0_ft_w
        # Range-check gains (ops.check_channels): which convs contract each fp32 tensor that enters the fp16-pair region.  The conditional of a level is
        # cat[key (64), tap_0 .. tap_3 (64 each)] (SRFlowNet_arch.py:122-137): tap k is read by rows 64(k+1)..64(k+2) of every level's two first convs,
        # a key tensor by rows 0..64 of its level's (fea_up0 is a bilinear resize of fea_up1: its level counts as a reader of fea_up1) and by the next
        # upconv.
        self._tap_gain, self._key_gain = {}, {}
        if self.concat and self.block_idxs and lvl_w:
            for k in range(len(self.block_idxs)):
                self._tap_gain[k] = _gain(ops, *[(w, 64 * (k + 1), 64 * (k + 2)) for ws_ in lvl_w.values() for w in ws_ if w.shape[1] >= 64 * (k + 2)])
        readers = {}
        for level, ws_ in lvl_w.items():
            name = self.level_names[level]
            readers.setdefault("fea_up1" if name == "fea_up0" else name, []).extend((w, 0, 64) for w in ws_)
        prev_name = "fea_up1"
        for kname, n in chain[:deepest + 1]:
            readers.setdefault(prev_name, []).append((g(n + ".weight"), 0, 64))
            prev_name = kname
        for name, cv in readers.items():
            self._key_gain[name] = _gain(ops, *cv)
        self.rrdb.out_gain = self._key_gain.get("fea_up1")

    def _level_shift(self, level):
        return _KEY_SHIFT.get(self.level_names[level])

    def _lr_level(self):
        """The flow level whose conditional lives at LR resolution (holds the block taps un-resized), or None."""
        for l in range(1, self.L + 1):
            if self._level_shift(l) == 0:
                return l
        return None

    def _taps_up2(self, level):
        """shift (1 = x2, 2 = x4) if this level's stacked RRDB taps are the LR-resolution taps nearest-upsampled by 2^shift and
        can be consumed in place through a parity-decomposed conv kernel, else 0.  x4 exists on the 3xBF16 path only."""
        sh = self._level_shift(level)
        if not (self.concat and self.block_idxs and self._lr_level() is not None):
            return 0
        if sh == 1 or (sh == 2 and getattr(self.ops, "conv_mode", "f32") == "x3"):
            return sh
        return 0

    # ------------------------------------------------------------------------------------------
    def _level_hw(self, level, h, w):
        """spatial size of flow level `level` for an LR of h x w: HR / 2^level."""
        return (h * self.scale) >> level, (w * self.scale) >> level

    def conditioning(self, lr, reverse=False, quads=True):
        """RRDB features + hoisted ft-only coupling activations for an LR batch (cached per tensor).
        reverse: the caller walks the levels from L down to 1 (decode), which decides which level is needed first.
        quads: hoisted tensors that only the coupling_head / coupling_tail pair reads are written QUAD-MAJOR ([B][C/4][H][W][4] in the
        same buffer: 16-byte accesses on both sides; `pre_fmt`, `h_ft_fmt` of the conditioning entry say which).  False (the likelihood
        path, which runs the generic kernels) keeps every tensor NCHW."""
        # the cache holds a reference to the keyed tensor, so its storage cannot be recycled for another input
        # while the entry is alive; a new tensor object or an in-place update (version bump) is a miss
        quads = bool(quads) and _QUADS
        key = (lr, lr._version, quads)
        if self._cond_key is not None and self._cond_key[0] is lr and self._cond_key[1] == lr._version and self._cond_key[2] == quads:
            return self._cond
        ops, ws = self.ops, self.ws
        if self._side_stream is not None:
            # hoists of the previous conditioning may still be in flight on the side stream (e.g. a level that was never
            # consumed after an exception): they write the workspaces this call is about to rewrite
            torch.cuda.current_stream(ops.device).wait_stream(self._side_stream)
        B, _, h, w = lr.shape
        if (h * self.scale) % (1 << self.L) or (w * self.scale) % (1 << self.L):
            raise ValueError("HR size must be divisible by 2^L (LR %dx%d, scale %d, L %d)" % (h, w, self.scale, self.L))
        ft = {}
        for level in range(1, self.L + 1):
            hl, wl = self._level_hw(level, h, w)
            # levels whose taps are consumed through conv_up2 only keep their 64 key channels
            ft[level] = ws.get("ft%d" % level, B, 64 if self._taps_up2(level) else self.n_cond, hl, wl)
        name2level = {self.level_names[l]: l for l in range(1, self.L + 1)}

        def key_view(name):
            if name in name2level:
                return ft[name2level[name]][:, :64]
            s = _KEY_SHIFT[name]
            return ws.get("key_" + name, B, 64, h << s if s >= 0 else h >> -s, w << s if s >= 0 else w >> -s)

        def on_block(idx, fea, b0, b1):
            # nearest-resize the tapped RRDB output (samples b0..b1) into its 64-ch slot of every level (SRFlowNet_arch.py:122-137)
            if idx in self.block_idxs and self.concat:
                k = self.block_idxs.index(idx)
                _chk(ops, fea, self._tap_gain.get(k))
                for level in range(1, self.L + 1):
                    if self._taps_up2(level):
                        continue
                    dst = ft[level][b0:b1, 64 * (k + 1): 64 * (k + 2)]
                    ops.resize(fea, dst, MODE_NEAREST, float(h) / dst.shape[2], float(w) / dst.shape[3])

        last = key_view("fea_up1")
        # second stream for the split-batch dense blocks (see RRDB._forward_packed); BFSR_OVERLAP=0 keeps everything on one stream
        if getattr(getattr(ops, "device", None), "type", "cpu") == "cuda" and os.environ.get("BFSR_OVERLAP", "auto") != "0":
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=ops.device)
            self.rrdb.side_stream = self._side_stream
        else:
            self.rrdb.side_stream = None
        self.rrdb.forward(lr, last, on_block, taps=set(self.block_idxs) if self.concat else set())
        prev = last
        for name in ("fea_up2", "fea_up4", "fea_up8"):       # lrelu is in-place in the reference => stored post-act
            if name in self.upconvs:
                cur = key_view(name)
                self.upconvs[name].run(ops, prev, cur, in_shift=1, act=ACT_LRELU, slope=0.2)
                _chk(ops, cur, self._key_gain.get(name))
                prev = cur
        if "fea_up0" in self.need_keys:
            dst = key_view("fea_up0")       # bilinear 1/2, align_corners=False, recompute_scale_factor=True
            ops.resize(last, dst, MODE_BILINEAR, float(h) / dst.shape[2], float(w) / dst.shape[3])

        # Only the level the caller starts with is needed right away (level 1 for encode, level L for decode); its steps are
        # short kernels with the matrix pipe ~40 % busy and leave room on the chip, so the hoisted convs of the other levels go
        # to a side stream that forks after the first level's hoists have been enqueued and joins at the first use of a level
        # (`_await`).  BFSR_OVERLAP=0 disables it.
        # ... unless the batch already fills the chip: at BASELINE config 4 (64 crops of 96x96 on one GPU) the side stream only adds
        # contention (867 -> 848 ms per step without it, profiles/r03_cfg4_bench_no_overlap.json; config 2, 8 x 160x160: 125 -> 121 ms
        # WITH it).  BFSR_OVERLAP=0 / 1 forces either way.
        ov = os.environ.get("BFSR_OVERLAP", "auto")
        use_side = (getattr(getattr(ops, "device", None), "type", "cpu") == "cuda"
                    and (ov == "1" or (ov not in ("0",) and B * h * w <= 300000)))
        main_stream = torch.cuda.current_stream(ops.device) if use_side else None
        cond = {}
        order = sorted(self.hoist.items(), reverse=bool(reverse))
        for level, hz in order:
            self._hoist_buffers(level, hz, ft, B)       # allocate on the main stream (see _hoist_buffers)
        for n, (level, hz) in enumerate(order):
            side = None
            if use_side and n >= 1:
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=ops.device)
                side = self._side_stream
                side.wait_stream(main_stream)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                cond[level] = self._hoist_level(level, hz, ft, B, quads)
                if side is not None:
                    cond[level]["ready"], cond[level]["joined"] = side.record_event(), set()
        self._cond_key, self._cond = key, cond
        return cond

    def _await(self, cnd):
        """Join the side stream that produced this level's hoisted tensors.  Idempotent per CONSUMER stream: the event is kept
        (the conditioning is cached across encode/decode and may be consumed from another torch stream later), and a stream
        that has already waited is remembered so the wait is enqueued once per stream."""
        ev = cnd.get("ready")
        if ev is not None:
            st = torch.cuda.current_stream(self.ops.device)
            if st.cuda_stream not in cnd["joined"]:
                st.wait_event(ev)
                cnd["joined"].add(st.cuda_stream)
        return cnd

    def _hoist_buffers(self, level, hz, ft, B):
        """The workspace tensors of a level's hoists (allocated on the CALLER's stream: buffers that are later filled on the
        side stream must not be owned by it, or the caching allocator could recycle them without a join)."""
        ws = self.ws
        K = len(hz["idxs"])
        hl, wl = ft[level].shape[2], ft[level].shape[3]
        Cz = [ly.C for ly in self.layers if ly.index == hz["idxs"][0]][0]
        if hz.get("x3s"):                                   # x3 copy of the channels the x3s hoists read (packed 16-bit: not a _Workspace)
            cx = 64 if hz["up2"] else ft[level].shape[1]
            key = (B, cx, hl, wl)
            if getattr(self, "_ftx3", None) is None:
                self._ftx3 = {}
            if level not in self._ftx3 or self._ftx3[level][0] != key:
                self._ftx3[level] = (key, self.ops.x3_empty(B, cx, hl, wl))
        # h2 tensors the level's hoists fill (possibly on the side stream): allocated HERE, per level, like every other hoist buffer
        if hz.get("ffast"):
            h2 = self._hid.get("ffh%d" % level)
            if h2 is None or tuple(h2.shape) != (B, 8, 2, hl, wl, 8):
                self._hid["ffh%d" % level] = self.ops.h2_empty(B, 64, hl, wl)
        if hz.get("h4t") is not None and os.environ.get("BFSR_UP4C", "1") == "1":      # compact taps result of the x4 level (see _hoist_level)
            tp = ft[self._lr_level()]
            ws.get("up4c%d" % level, B, 9 * K * 64, tp.shape[2], tp.shape[3])
        if hz.get("up2") and (hz.get("h2t") is not None or hz.get("h4t") is not None):
            taps = ft[self._lr_level()][:, 64:]
            key = (B, taps.shape[1] + (4 * 64 if hz.get("h2t") is not None else 0)) + tuple(taps.shape[2:])
            if getattr(self, "_taps_h2", None) is None:
                self._taps_h2 = {}
            if level not in self._taps_h2 or self._taps_h2[level][0] != key:
                self._taps_h2[level] = (key, self.ops.h2_empty(*key))
        if (not hz.get("up2") and getattr(self.ops, "conv_mode", "f32") == "x3" and getattr(self.ops, "split", "") == "f16x2"
                and hasattr(self.ops, "conv_h2x") and ft[level].shape[1] % 16 == 0 and os.environ.get("BFSR_HOIST", "h2x") == "h2x"
                and hz.get("aff0") is not None and getattr(hz["aff0"], "mode", "") == "x3"):
            key = (B, ft[level].shape[1], hl, wl)
            if getattr(self, "_ft_h2", None) is None:
                self._ft_h2 = {}
            if level not in self._ft_h2 or self._ft_h2[level][0] != key:
                self._ft_h2[level] = (key, self.ops.h2_empty(*key))
        elif getattr(self, "_ft_h2", None):
            self._ft_h2.pop(level, None)
        # levels whose coupled steps all run the unfused h2 chain (C = 96): pre_aff only ever enters a conv_h2x launch as its residual -> h2 tensor
        if getattr(self, "_pre_h2", None) is None:
            self._pre_h2 = {}
        if (level in getattr(self, "_ft_h2", {}) and os.environ.get("BFSR_L3", "h2x") == "h2x"
                and all(getattr(self.steps[i], "chain", None) is not None for i in hz["idxs"])):
            key = (B, K * 64, hl, wl)
            if level not in self._pre_h2 or self._pre_h2[level][0] != key:
                self._pre_h2[level] = (key, self.ops.h2_empty(*key))
        else:
            self._pre_h2.pop(level, None)
        return (ws.get("hoist_hid%d" % level, B, K * 64, hl, wl), ws.get("pre_aff%d" % level, B, K * 64, hl, wl),
                ws.get("h_ft%d" % level, B, K * 2 * Cz, hl, wl), Cz)

    def _hoist_level(self, level, hz, ft, B, quads=False):
        ops = self.ops
        f = ft[level]
        hid, pre_aff, h_ft, Cz = self._hoist_buffers(level, hz, ft, B)
        pre_h2 = None
        # quad-major hand-over to the coupling pair: pre_aff as a whole (one batched conv writes it) when every step of the level is
        # fused and its producers can (the register-staged x4 taps kernel cannot, conv_up4_h2t can); h_ft per step (each step's Conv2dZeros writes its own slice) when the
        # step's h_ft is only ever read by coupling_tail -- i.e. not by flow_pointwise as the first step of an encode pass
        pq = int(quads and hz.get("pre_q4", False))
        hq = {i: int(quads and (i in hz.get("hft_q4", ()) or (i in hz.get("hft_q4_wide", ()) and level in self._pre_h2))) for i in hz["idxs"]}
        kq = dict(y_fmt=1) if pq else {}
        ff = bool(hz.get("ffast")) and hz.get("x3", True) is not False
        h4t = hz.get("h4t") if (ff and pq and hz.get("x3s") and hz.get("up2") and hz.get("up") == 2) else None
        ffq = int(ff and not (hz["up2"] and hz.get("up") == 2 and h4t is None))     # layout of the raw fFeatures.0 result the 1x1-only head reads: 1 quad-major, 0 NCHW
        if h4t is not None:
            # x4 level on conv_up4_h2t: the key convs (channels at output resolution) write quad-major, the taps kernel adds its result in place
            taps = ft[self._lr_level()][:, 64:]
            taps_h2 = ops.h2_pack(taps, self._taps_h2[level][1])
            f3 = ops.x3_pack(f, self._ftx3[level][1])
            if os.environ.get("BFSR_UP4C", "1") == "1":
                # DEFAULT since round 6: the taps kernel writes its nine class values per source pixel (9/16 of the full-resolution bytes, no
                # read-back of the key conv's result) and the key conv expands and adds them while it writes the full-resolution tensor; bit-identical
                # to the pre_add form (tests/test_srflow_gpu.py::test_x4_level_compact_taps_equals_pre_add).  Config 4 (64 x 96^2): 432.9 -> 424.6 ms
                # interleaved on one box (profiles/r06p_ab_up4c.txt); the isolated pair of launches at 16 x 96^2 had measured equal (18.1 vs 18.5 ms per
                # tensor), the gain shows where the 38.7 GB read-back competes with the rest of the pass.  Costs a 0.34 GB-per-crop buffer (21.7 GB at 64).
                comp = self.ws.get("up4c%d" % level, B, 9 * hid.shape[1], taps.shape[2], taps.shape[3])
                ops.conv_up4_h2t(taps_h2, h4t[0], comp, compact=True)
                ops.conv_x3s(f3, hz["ft0_key"], hid, y_fmt=1, up4=comp)
                ops.conv_up4_h2t(taps_h2, h4t[1], comp, compact=True)
                ops.conv_x3s(f3, hz["aff0_key"], pre_aff, y_fmt=1, up4=comp)
            else:
                ops.conv_x3s(f3, hz["ft0_key"], hid, y_fmt=1)
                ops.conv_up4_h2t(taps_h2, h4t[0], hid, pre_add=hid)
                ops.conv_x3s(f3, hz["aff0_key"], pre_aff, y_fmt=1)
                ops.conv_up4_h2t(taps_h2, h4t[1], pre_aff, pre_add=pre_aff)
        elif hz["up2"]:
            taps = ft[self._lr_level()][:, 64:]
            if hz["x3"]:
                up = ops.conv_up4_x3 if hz["up"] == 2 else ops.conv_up2_x3
                fq = dict(y_fmt=1) if ffq else {}                      # ffast: raw conv result, quad-major where the kernels can (the 1x1-only head applies ActNorm + ReLU)
                h2t = hz.get("h2t") if (ff and pq and hz["x3s"]) else None
                if h2t is not None:
                    # one h2 tensor at LR resolution: the 256 tap channels, then the four space-to-depth planes of the 64 key channels
                    # (fea_up2) -- the key conv and the pre_add round trip are K chunks of the taps kernel
                    ct = taps.shape[1]
                    taps_h2 = self._taps_h2[level][1]                  # allocated by _hoist_buffers on the caller's stream
                    ops.h2_pack(taps, taps_h2[:, :ct // 8])
                    ops.h2_pack_s2d(f[:, :64], taps_h2[:, ct // 8:])
                    ops.conv_up2_h2t(taps_h2, h2t[0], hid)
                    ops.conv_up2_h2t(taps_h2, h2t[1], pre_aff)
                if h2t is not None:
                    pass
                elif hz["x3s"]:
                    f3 = ops.x3_pack(f, self._ftx3[level][1])
                    ops.conv_x3s(f3, hz["ft0_key"], hid, **fq)
                else:
                    ops.conv_x3(f, hz["ft0_key"], hid, **fq)
                if h2t is not None:
                    pass
                elif ff:
                    up(taps, hz["ft0_taps"], hid, pre_add=hid, **fq)
                else:
                    up(taps, hz["ft0_taps"], hid, epi=hz["ft0_epi"], act=ACT_RELU, pre_add=hid)
                if h2t is not None:
                    pass
                elif hz["x3s"]:
                    ops.conv_x3s(f3, hz["aff0_key"], pre_aff, **kq)
                else:
                    ops.conv_x3(f, hz["aff0_key"], pre_aff, **kq)
                if h2t is None:
                    up(taps, hz["aff0_taps"], pre_aff, pre_add=pre_aff, **kq)
            else:
                ops.conv_up2(taps, hz["ft0_taps"], hid, epi=hz["ft0_epi"], act=ACT_RELU, key=(f, hz["ft0_key"]))
                ops.conv_up2(taps, hz["aff0_taps"], pre_aff, key=(f, hz["aff0_key"]))
        elif level in getattr(self, "_ft_h2", {}):
            # levels 2 / 3 (the conditional IS the stacked features, no upsampling): the two batched 320 -> 16*64 hoists on the LDS-DMA kernel
            # conv_h2x over ONE h2 copy of the level's features instead of the register-staged split conv (round 5; measured on the round-4
            # build, tools/exp/hoist2_bench.py: 3.38 -> 0.09 pack + 3.13 ms at 8 x 160^2, 1.13 -> 0.02 + 0.98 ms at 8 x 80^2 per conv)
            fh = ops.h2_pack(f, self._ft_h2[level][1])
            if ff:
                ops.conv_h2x(fh, hz["ft0_raw"].pw, hid, epi=hz["ft0_raw"].epi, y_fmt=1)
            else:
                ops.conv_h2x(fh, hz["ft0"].pw, hid, epi=hz["ft0"].epi, act=ACT_RELU)
            if level in self._pre_h2:
                pre_h2 = ops.conv_h2x(fh, hz["aff0"].pw, self._pre_h2[level][1], epi=hz["aff0"].epi)
            else:
                ops.conv_h2x(fh, hz["aff0"].pw, pre_aff, epi=hz["aff0"].epi, **kq)
        else:
            if ff:
                hz["ft0_raw"].run(ops, f, hid, y_fmt=1)
            else:
                hz["ft0"].run(ops, f, hid, act=ACT_RELU)
            hz["aff0"].run(ops, f, pre_aff, **kq)
        for k, i in enumerate(hz["idxs"]):
            st = self.steps[i]
            hk = hid[:, 64 * k: 64 * (k + 1)]
            if ff:
                h2 = self._hid["ffh%d" % level]                       # allocated by _hoist_buffers on the caller's stream
                ops.coupling_head(None, st.fthead, hk, h2, pre_fmt=ffq)
                if st.ft4x is not None:
                    ops.conv_h2x(h2, st.ft4x[0], h_ft[:, 2 * Cz * k: 2 * Cz * (k + 1)], epi=st.ft4x[1], y_fmt=hq[i])
                    continue
                for pk, epi, g0, g1 in st.ft4r:
                    ops.conv_h2r(h2, pk, h_ft[:, 2 * Cz * k + g0: 2 * Cz * k + g1], epi=epi, y_fmt=hq[i])
                continue
            st.ft2.run(ops, hk, hk, act=ACT_RELU)            # 1x1, in place (disjoint pixel tiles)
            st.ft4.run(ops, hk, h_ft[:, 2 * Cz * k: 2 * Cz * (k + 1)], **(dict(y_fmt=1) if hq[i] else {}))
        return dict(pre_aff=pre_aff, h_ft=h_ft, slot={i: k for k, i in enumerate(hz["idxs"])}, C=Cz, pre_fmt=pq, h_ft_fmt=hq, pre_h2=pre_h2)

    # ------------------------------------------------------------------------------------------
    def _self_cond(self, st, z, cnd, k, tag):
        """h_aff = fAffine(cat[z1, ft]) with the ft rows hoisted: 3x3 on z1 (+pre_aff) -> 1x1 -> 3x3."""
        ops, ws = self.ops, self.ws
        B, C, H, W = z.shape
        cn = C // 2
        h_aff = ws.get("haff_%s_%d" % (tag, k & 1), B, 2 * (C - cn), H, W)
        if getattr(st, "chain", None) is not None and not cnd.get("pre_fmt"):
            p0, hp, p4, e4 = st.chain
            raw = ws.get("raw_%s" % tag, B, 64, H, W)
            if cnd.get("pre_h2") is not None:
                # round 5: the split 3x3 on z1 on the LDS-DMA kernel too (z1 packed into an h2 tensor, the hoisted partial = its h2 residual):
                # the register-staged kernel ran this 48 -> 64 conv at 0.15 PFLOP/s (660 us per step at 64 x 96^2, 45 us at 8 x 80^2)
                zk = "z1h_" + tag
                zh = self._hid.get(zk)
                if zh is None or tuple(zh.shape) != (B, cn // 8, 2, H, W, 8):
                    zh = self._hid[zk] = ops.h2_empty(B, cn, H, W)
                ops.h2_pack(z[:, :cn], zh)
                ops.conv_h2x(zh, p0, raw, res1=cnd["pre_h2"][:, 8 * k: 8 * (k + 1)], alpha1=1.0)
            else:
                ops.conv_x3(z[:, :cn], p0, raw, pre_add=cnd["pre_aff"][:, 64 * k: 64 * (k + 1)])
            key = "hidc_" + tag
            h2 = self._hid.get(key)
            if h2 is None or tuple(h2.shape) != (B, 8, 2, H, W, 8):
                h2 = self._hid[key] = ops.h2_empty(B, 64, H, W)
            ops.coupling_head(None, hp, raw, h2, pre_fmt=0)
            ops.conv_h2x(h2, p4, h_aff, epi=e4)
            return h_aff
        hid = ws.get("hid_%s" % tag, B, 64, H, W)
        # 3x3 on z1 (+ hoisted ft partial, ActNorm, ReLU) with the 1x1 (+ActNorm, ReLU) fused as a second MFMA stage
        st.aff0_z1.run(ops, z[:, :cn], hid, pre_add=cnd["pre_aff"][:, 64 * k: 64 * (k + 1)], act=ACT_RELU,
                       stage2=(st.aff2.pw, st.aff2.epi, ACT_RELU))
        st.aff4.run(ops, hid, h_aff)
        return h_aff

    def _pairs(self, st, cnd, logdet=None):
        """Does this coupled step run as a head / tail kernel pair (levels 1 / 2: coupling.hip + coupling_tail.hip; the wide level: coupling_wide.hip)?"""
        if logdet is not None:
            return False
        return bool(getattr(st, "fused", False)) or (getattr(st, "wide", None) is not None and cnd.get("pre_h2") is not None)

    def _wide_info(self, st, cnd, k, ly, nxt):
        """What the wide pair needs beyond the pair's arguments: the h2 view of the hoisted partial, this step's layer index and -- when the step that
        runs next on this z is another wide step -- that step's index (the tail then writes its z1 as an h2 tensor)."""
        if getattr(st, "wide", None) is None or cnd.get("pre_h2") is None:
            return None
        nx = None
        if nxt is not None and nxt.type == "step" and nxt.coupled and nxt.level == ly.level and getattr(self.steps[nxt.index], "wide", None) is not None:
            nx = nxt.index
        return dict(pre_h2=cnd["pre_h2"][:, 8 * k: 8 * (k + 1)], idx=ly.index, nxt=nx)

    def _pair(self, st, z, pre_k, tag, reverse, kw, pre_fmt=0, wide=None):
        """coupling_head -> coupling_tail, in place on z; `hid` travels between them as an h2 tensor (fp16 hi + lo planes)."""
        ops = self.ops
        B, _, H, W = z.shape
        if wide is not None:
            cn = st.wide["Cz"]
            zk = "z1h_" + tag
            zh = self._hid.get(zk)
            if zh is None or tuple(zh.shape) != (B, cn // 8, 2, H, W, 8):
                zh = self._hid[zk] = ops.h2_empty(B, cn, H, W)
                self._z1h_next.pop(tag, None)
            if self._z1h_next.pop(tag, None) != (z.data_ptr(), wide["idx"]):
                ops.h2_pack(z[:, :cn], zh)                  # first wide step of a run: the previous kernel on z was not a wide tail
            hk = "hidc_" + tag
            hid = self._hid.get(hk)
            if hid is None or tuple(hid.shape) != (B, 8, 2, H, W, 8):
                hid = self._hid[hk] = ops.h2_empty(B, 64, H, W)
            ops.coupling_wide_head(zh, st.wide, wide["pre_h2"], hid)
            kw = dict(kw)
            if kw.get("w") is not None:
                kw["w"] = self._wide_w[kw["w"].data_ptr()]
            ops.coupling_wide_tail(hid, st.wide, z, z, reverse, z1h=zh if wide["nxt"] is not None else None, **kw)
            if wide["nxt"] is not None:
                self._z1h_next[tag] = (z.data_ptr(), wide["nxt"])
            return z
        key = "hid_" + tag
        hid = self._hid.get(key)
        if hid is None or tuple(hid.shape) != (B, 8, 2, H, W, 8):
            hid = self._hid[key] = ops.h2_empty(B, 64, H, W)
        ops.coupling_head(z, st.head, pre_k, hid, pre_fmt=pre_fmt)
        return ops.coupling_tail(hid, st.tail, z, z, reverse, **kw)

    # ---- two half-batch lanes for the short levels of the chain --------------------------------------------------------------------
    def _lane_run(self, pos, step, B, H, W, logdet):
        """The maximal run of consecutive step layers of one level starting at self.layers[pos] (walking in direction `step`), if it should be
        executed as two half-batch lanes on two streams.  Level 3 (C = 96): four short launches per coupled step (split 3x3, 1x1-only head,
        Conv2dZeros, pointwise) that fill < 1/4 of the chip at config 2 (8 x 80^2: 120 tiles) and are bound by their own latency chain: two
        independent half batches in flight overlap those latencies (every kernel is per-sample, so the results are
        bit-identical; -0.6 ms per config-2 step).  The generators below also handle the fused pair (levels 1 / 2): measured with level 2 of
        config 2 included (8 x 160^2, 400 tiles per launch) the gain is the same 0.5-0.6 ms, i.e. level 2 adds nothing, so the bound stays at
        100 000 pixels per level -- larger levels keep one lane; small inputs run every level in lanes (covered by the B = 2 tests).
        None when the lanes would not help or cannot be used."""
        if (logdet is not None or B < 2 or B % 2 or B * H * W > 100000 or os.environ.get("BFSR_LANES", "1") == "0"
                or os.environ.get("BFSR_OVERLAP", "auto") == "0" or getattr(getattr(self.ops, "device", None), "type", "cpu") != "cuda"):
            return None
        run, p, level = [], pos, self.layers[pos].level
        while 0 <= p < len(self.layers) and self.layers[p].type == "step" and self.layers[p].level == level:
            run.append(self.layers[p])
            p += step
        return run if len(run) >= 4 else None

    def _lanes(self, make_gen, B):
        """Run make_gen(b0, b1, lane) for the two halves of the batch, enqueued alternately on the current and the lane stream."""
        ops = self.ops
        main = torch.cuda.current_stream(ops.device)
        if getattr(self, "_lane_stream", None) is None:         # its own stream: the side stream may be busy with the prior's branch 0 or the hoists
            self._lane_stream = torch.cuda.Stream(device=ops.device)
        side = self._lane_stream
        ga, gb = make_gen(0, B // 2, 0), make_gen(B // 2, B, 1)
        side.wait_stream(main)
        live = True
        while live:
            live = next(ga, False) is not False
            with torch.cuda.stream(side):
                live = (next(gb, False) is not False) or live
        main.wait_stream(side)

    def _lane_cond(self, cnd, b0, b1):
        c = dict(cnd)
        c["pre_aff"], c["h_ft"] = cnd["pre_aff"][b0:b1], cnd["h_ft"][b0:b1]
        if cnd.get("pre_h2") is not None:
            c["pre_h2"] = cnd["pre_h2"][b0:b1]
        return c

    def _steps_fwd_lane(self, run, z, cond, b0, b1, lane):
        """encode()'s step sequence of one level (its `pending` / `head_done` protocol) on the samples b0..b1; yields after every group of launches."""
        ops = self.ops
        zl, pending, head_done = z[b0:b1], None, False
        for n, ly in enumerate(run):
            st = self.steps[ly.index]
            tag = "enc%d_l%d" % (ly.level, lane)
            if ly.coupled:
                cnd = self._lane_cond(self._await(cond[ly.level]), b0, b1)
                k = cnd["slot"][ly.index]
                if not head_done:
                    ops.flow_pointwise(zl, zl, False, h_aff=pending, an_bias=st.an_bias, an_escale=st.an_exp, w=st.w_fwd, wt=st.w_fwd_t,
                                       h_ft=cnd["h_ft"][:, 2 * ly.C * k: 2 * ly.C * (k + 1)])
                    pending = None
                    yield
                head_done = False
                if self._pairs(st, cnd):
                    nxt = run[n + 1] if n + 1 < len(run) else None      # (the layer behind a whole-level run is never a step)
                    kw = {}
                    if nxt is not None:
                        sn = self.steps[nxt.index]
                        kw = dict(an_bias=sn.an_bias, an_escale=sn.an_exp, w=sn.w_fwd)
                        if nxt.coupled:
                            kn = cnd["slot"][nxt.index]
                            kw["h_ft"] = cnd["h_ft"][:, 2 * ly.C * kn: 2 * ly.C * (kn + 1)]
                            kw["h_ft_fmt"] = cnd["h_ft_fmt"][nxt.index]
                        head_done = True
                    self._pair(st, zl, cnd["pre_aff"][:, 64 * k: 64 * (k + 1)], tag, False, kw, cnd["pre_fmt"], wide=self._wide_info(st, cnd, k, ly, nxt))
                else:
                    pending = self._self_cond(st, zl, cnd, k, tag)
                yield
            else:
                if not head_done:
                    ops.flow_pointwise(zl, zl, False, h_aff=pending, an_bias=st.an_bias, an_escale=st.an_exp, w=st.w_fwd, wt=st.w_fwd_t)
                    yield
                pending, head_done = None, False
        if pending is not None:
            ops.flow_pointwise(zl, zl, False, h_aff=pending)
            yield

    def _steps_rev_lane(self, run, z, cond, b0, b1, lane):
        """decode()'s step sequence of one level on the samples b0..b1."""
        ops = self.ops
        zl = z[b0:b1]
        C = zl.shape[1]
        for n, ly in enumerate(run):
            st = self.steps[ly.index]
            tag = "dec%d_l%d" % (ly.level, lane)
            if ly.coupled:
                cnd = self._lane_cond(self._await(cond[ly.level]), b0, b1)
                k = cnd["slot"][ly.index]
                if self._pairs(st, cnd):
                    kw = dict(h_ft=cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)], h_ft_fmt=cnd["h_ft_fmt"][ly.index], w=st.w_inv,
                              an_bias=st.an_bias, an_escale=st.an_expneg)
                    nxt = run[n + 1] if n + 1 < len(run) else None      # the step that runs next (the run walks the layers backwards)
                    self._pair(st, zl, cnd["pre_aff"][:, 64 * k: 64 * (k + 1)], tag, True, kw, cnd["pre_fmt"], wide=self._wide_info(st, cnd, k, ly, nxt))
                    yield
                else:
                    h_aff = self._self_cond(st, zl, cnd, k, tag)
                    yield
                    ops.flow_pointwise(zl, zl, True, h_aff=h_aff, h_ft=cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)],
                                       w=st.w_inv, wt=st.w_inv_t, an_bias=st.an_bias, an_escale=st.an_expneg)
                    yield
            else:
                ops.flow_pointwise(zl, zl, True, w=st.w_inv, wt=st.w_inv_t, an_bias=st.an_bias, an_escale=st.an_expneg)
                yield

    def encode(self, gt, lr, logdet=None, on_eps=None):
        """normal flow (FlowUpsamplerNet.encode :217-251): gt [B,3,H,W] -> [eps_split..., z_final].
        logdet: optional float64 [B] accumulator that receives the flow's log-determinant (actnorm + invconv constants,
        sum log(scale) of both couplings per step, Split2d log-likelihood) -- the reference's `logdet` return value.
        on_eps(i, eps_i): called as soon as the split that produces latent i has been enqueued (the LP harness starts that latent's branch of
        the prior on the side stream while the remaining levels of the chain run, srflow/test.py)."""
        ops, ws = self.ops, self.ws
        cond = self.conditioning(lr, quads=logdet is None)
        self._z1h_next.clear()      # (a pass that was interrupted must not leave a "z1 copy is valid" note behind)
        z = gt
        epses = []
        pending = None           # h_aff of the previous coupled step, applied lazily by the next head
        ld_const, ld_levels = 0.0, set()
        head_done = False        # the current step's head (actnorm, W, feature-conditional affine) was already applied by the
                                 # previous step's fused tail kernel
        skip_to = -1
        for pos, ly in enumerate(self.layers):
            if pos < skip_to:
                continue
            B, _, H, W = z.shape
            if ly.type == "step" and pending is None and not head_done:
                run = self._lane_run(pos, +1, B, H, W, logdet)
                if run is not None:
                    self._lanes(lambda b0, b1, lane, run=run, z=z: self._steps_fwd_lane(run, z, cond, b0, b1, lane), B)
                    skip_to = pos + len(run)
                    continue
            if ly.type == "squeeze":
                if pending is not None:
                    ops.flow_pointwise(z, z, False, h_aff=pending)
                    pending = None
                out = ws.get("enc_z%d" % ly.level, B, ly.C, H // 2, W // 2)
                z = _chk(ops, ops.squeeze2d(z, out), tiny=0.0)        # the coupling heads split z1 into fp16 pairs (overflow side only: see decode)
            elif ly.type == "step":
                st = self.steps[ly.index]
                if ly.coupled:
                    cnd = self._await(cond[ly.level])
                    k = cnd["slot"][ly.index]
                    if not head_done:
                        ops.flow_pointwise(z, z, False, h_aff=pending, an_bias=st.an_bias, an_escale=st.an_exp,
                                           w=st.w_fwd, wt=st.w_fwd_t, h_ft=cnd["h_ft"][:, 2 * ly.C * k: 2 * ly.C * (k + 1)])
                        pending = None
                    head_done = False
                    if self._pairs(st, cnd, logdet):
                        # the tail applies this step's self-conditional affine and, when the next layer is another step of this
                        # level, that step's head (what the generic path does lazily through `pending`)
                        nxt = self.layers[pos + 1] if pos + 1 < len(self.layers) else None
                        kw = {}
                        if nxt is not None and nxt.type == "step":
                            sn = self.steps[nxt.index]
                            kw = dict(an_bias=sn.an_bias, an_escale=sn.an_exp, w=sn.w_fwd)
                            if nxt.coupled:
                                kn = cnd["slot"][nxt.index]
                                kw["h_ft"] = cnd["h_ft"][:, 2 * ly.C * kn: 2 * ly.C * (kn + 1)]
                                kw["h_ft_fmt"] = cnd["h_ft_fmt"][nxt.index]
                            head_done = True
                        pre_k = cnd["pre_aff"][:, 64 * k: 64 * (k + 1)]
                        z = self._pair(st, z, pre_k, "enc%d" % ly.level, False, kw, cnd["pre_fmt"], wide=self._wide_info(st, cnd, k, ly, nxt))
                    else:
                        pending = self._self_cond(st, z, cnd, k, "enc%d" % ly.level)
                        if logdet is not None:
                            ops.logscale_sum(pending, logdet, 1.0)
                            if ly.level not in ld_levels:           # the hoisted h_ft holds the scaleFt of all K steps of the level
                                ld_levels.add(ly.level)
                                ops.logscale_sum(cnd["h_ft"], logdet, 1.0)
                else:
                    if not head_done:
                        ops.flow_pointwise(z, z, False, h_aff=pending, an_bias=st.an_bias, an_escale=st.an_exp, w=st.w_fwd, wt=st.w_fwd_t)
                    pending, head_done = None, False
                ld_const += st.ld_const * H * W
            else:   # split
                if pending is not None:
                    ops.flow_pointwise(z, z, False, h_aff=pending)
                    pending = None
                h = ws.get("split_h%d" % ly.index, B, 2 * ly.C_consume, H, W)
                self.splits[ly.index].run(ops, z[:, :ly.C_pass], h)
                e = ops.empty(B, ly.C_consume, H, W)
                ops.split2d(h, z[:, ly.C_pass:], e, False)
                if logdet is not None:
                    ops.gaussian_logp(z[:, ly.C_pass:], logdet, h=h, coef=1.0)
                epses.append(e)
                if on_eps is not None:
                    on_eps(len(epses) - 1, e)
                z = z[:, :ly.C_pass]
        if pending is not None:
            ops.flow_pointwise(z, z, False, h_aff=pending)
        zf = ops.empty(*z.shape)
        ops.axpb_clamp(z, zf)               # detach the result from the workspace
        epses.append(zf)
        if logdet is not None:
            logdet += ld_const
        return epses

    def decode(self, lr, epses=None, z=None, eps_std=None, logdet=None, eps_ready=None):
        """reverse flow (FlowUpsamplerNet.decode :267-296): [eps_split..., z_final] -> sr [B,3,H,W].
        logdet: optional float64 [B] accumulator (every term of encode() enters with the opposite sign).
        eps_ready: optional {index into epses: event}: the current stream waits for the event right before that latent is consumed."""
        ops, ws = self.ops, self.ws
        cond = self.conditioning(lr, reverse=True, quads=logdet is None)
        self._z1h_next.clear()
        ld_const, ld_levels = 0.0, set()
        epses = list(epses) if epses is not None else None
        zin = epses.pop() if epses is not None else z
        B = zin.shape[0]
        cur = ws.get("dec_z_top", *zin.shape)
        ops.axpb_clamp(zin, cur)
        # the coupling heads split z1 into fp16 pairs: checked for overflow only.  A tiny flow state (a dark crop) is harmless: the 3x3 conv on z1 is a
        # RESIDUAL on the hoisted partial pre_aff, so what its absolute split error (<= 2^-25 x the weight mass) is compared with is the hidden
        # pre-activation, whose size does not depend on z
        z = _chk(ops, cur, tiny=0.0)
        # the buffer the next (lower-index) split layer concatenates into is prepared when we reach a squeeze
        skip_to = len(self.layers)
        for pos in reversed(range(len(self.layers))):
            if pos > skip_to:
                continue
            ly = self.layers[pos]
            _, C, H, W = z.shape
            if ly.type == "step":
                run = self._lane_run(pos, -1, B, H, W, logdet)
                if run is not None:
                    self._lanes(lambda b0, b1, lane, run=run, z=z: self._steps_rev_lane(run, z, cond, b0, b1, lane), B)
                    skip_to = pos - len(run)
                    continue
            if ly.type == "step":
                st = self.steps[ly.index]
                if ly.coupled:
                    cnd = self._await(cond[ly.level])
                    k = cnd["slot"][ly.index]
                    if self._pairs(st, cnd, logdet):
                        pre_k = cnd["pre_aff"][:, 64 * k: 64 * (k + 1)]
                        kw = dict(h_ft=cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)], h_ft_fmt=cnd["h_ft_fmt"][ly.index], w=st.w_inv,
                                  an_bias=st.an_bias, an_escale=st.an_expneg)
                        z = self._pair(st, z, pre_k, "dec%d" % ly.level, True, kw, cnd["pre_fmt"],
                                       wide=self._wide_info(st, cnd, k, ly, self.layers[pos - 1] if pos > 0 else None))
                    else:
                        h_aff = self._self_cond(st, z, cnd, k, "dec%d" % ly.level)
                        if logdet is not None:
                            ops.logscale_sum(h_aff, logdet, -1.0)
                            if ly.level not in ld_levels:
                                ld_levels.add(ly.level)
                                ops.logscale_sum(cnd["h_ft"], logdet, -1.0)
                        ops.flow_pointwise(z, z, True, h_aff=h_aff, h_ft=cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)],
                                           w=st.w_inv, wt=st.w_inv_t, an_bias=st.an_bias, an_escale=st.an_expneg)
                else:
                    ops.flow_pointwise(z, z, True, w=st.w_inv, wt=st.w_inv_t, an_bias=st.an_bias, an_escale=st.an_expneg)
                ld_const -= st.ld_const * H * W
            elif ly.type == "squeeze":
                Co = C // 4
                nxt = self.layers[pos - 1] if pos > 0 else None
                if nxt is not None and nxt.type == "split":
                    full = ws.get("dec_full%d" % nxt.index, B, nxt.C, 2 * H, 2 * W)
                    out = full[:, :nxt.C_pass]
                    assert nxt.C_pass == Co
                elif pos == 0:
                    out = ops.empty(B, Co, 2 * H, 2 * W)
                else:
                    out = ws.get("dec_z%d" % ly.level, B, Co, 2 * H, 2 * W)
                z = ops.unsqueeze2d(z, out)
            else:   # split reverse (Split.py:62-76): z = cat(z1, mean + exp(logs)*eps)
                full = ws.get("dec_full%d" % ly.index, B, ly.C, H, W)
                assert z.data_ptr() == full.data_ptr()
                h = ws.get("split_h%d" % ly.index, B, 2 * ly.C_consume, H, W)
                self.splits[ly.index].run(ops, z, h)
                if epses is not None:
                    if eps_ready and (len(epses) - 1) in eps_ready:
                        torch.cuda.current_stream(ops.device).wait_event(eps_ready[len(epses) - 1])
                    e = epses.pop()
                else:   # tau path: eps ~ N(0, eps_std) sampled on device (plumbing)
                    e = rng.randn((B, ly.C_consume, H, W), z.device) * float(eps_std or 1)
                ops.split2d(h, e, full[:, ly.C_pass:], True)
                if logdet is not None:
                    ops.gaussian_logp(full[:, ly.C_pass:], logdet, h=h, coef=-1.0)
                z = full
        if logdet is not None:
            logdet += ld_const
        return z
