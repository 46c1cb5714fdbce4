"""Learned-prior UNets on the HIP kernels.

SRFlow-LP prior (SRFlow-LP/code/models/unet.py:109-181): two independent branches
  DenseBlock_5C(in -> 64) -> DoubleConv -> 3 x [MaxPool2, DoubleConv] -> 3 x [bilinear x2 (align_corners=True),
  pad, cat([skip, up]), DoubleConv(mid = in/2)] -> 1x1 conv.
BatchNorm runs in eval mode and is folded into the conv epilogue as (v - mean) * (gamma/sqrt(var+eps)) + beta.
The `cat([skip, up])` buffers are allocated up front: the encoder writes its skip features straight into
the first half, the upsampler into the second half.
"""
import os

import torch

from ..ops import ACT_LRELU, ACT_NONE, MODE_BILINEAR_AC
from .engine import _ConvP, _Workspace, h2_mode


def _min_tiles():
    """Tiles of 16 x 32 a SAMPLE must have for a UNet level to run on the LDS-DMA kernels (below it the register-staged split conv with its
    smaller tiles fills the chip better at small batches); BFSR_PRIOR_MIN_TILES overrides it for measurements."""
    return int(os.environ.get("BFSR_PRIOR_MIN_TILES", "32"))


def _bn_conv(ops, sd, wkey, bnp, eps=1e-5, f16=False):
    s = sd[bnp + ".weight"] / torch.sqrt(sd[bnp + ".running_var"] + eps)
    return _ConvP(ops, sd[wkey], aff_shift=-sd[bnp + ".running_mean"], aff_scale=s, aff_post=sd[bnp + ".bias"], f16=f16)


class _DoubleConv(object):
    def __init__(self, ops, sd, p, f16=False):
        self.c1 = _bn_conv(ops, sd, p + ".double_conv.0.weight", p + ".double_conv.1", f16=f16)
        self.c2 = _bn_conv(ops, sd, p + ".double_conv.3.weight", p + ".double_conv.4", f16=f16)
        self.mid = self.c1.pw.Cout
        self.out = self.c2.pw.Cout

    def run(self, ops, ws, tag, x, out):
        B, _, H, W = x.shape
        mid = ws.get(tag + "_mid", B, self.mid, H, W)
        self.c1.run(ops, x, mid, act=ACT_LRELU, slope=0.2)
        self.c2.run(ops, mid, out, act=ACT_LRELU, slope=0.2)
        return out

    def run_h2(self, ops, hb, tag, x_h2, out, lo=False):
        """Both convs on the LDS-DMA kernel of the contraction mode (_ConvP.run_h2): x_h2 an h2 view, `out` an h2 view or an fp32 tensor."""
        B, H, W = x_h2.shape[0], x_h2.shape[3], x_h2.shape[4]
        mid = hb(tag + "_mid_h2", "h2", B, self.mid, H, W)
        self.c1.run_h2(ops, x_h2, mid, act=ACT_LRELU, slope=0.2)
        self.c2.run_h2(ops, mid, out, act=ACT_LRELU, slope=0.2, lo=lo)
        return out


class DenseBlock(object):
    """DenseBlock_5C (unet.py:10-36): five 3x3 convs over a growing concat, no residual."""

    def __init__(self, ops, sd, p, f16=False):
        self.convs = [_ConvP(ops, sd["%s.conv%d.weight" % (p, i)], sd["%s.conv%d.bias" % (p, i)], f16=f16) for i in range(1, 6)]
        self._raw = [(sd["%s.conv%d.weight" % (p, i)], sd["%s.conv%d.bias" % (p, i)]) for i in range(1, 6)]
        self.f16 = f16
        self.nf = self.convs[0].pw.Cin
        self.gc = self.convs[0].pw.Cout
        self.out = self.convs[4].pw.Cout

    def h2_ready(self, ops):
        """Padded convs for the h2 path (run_h2): the block's nf input channels padded to the K chunk of the kernel family (16 channels for
        conv_h2x, 32 for conv_h2s) with zero weights."""
        if getattr(self, "_h2", None) is None:
            q = 32 if self.f16 else 16
            nfp = -(-self.nf // q) * q
            cs = []
            for i, (w, b) in enumerate(self._raw):
                w = w.detach().to("cpu", torch.float32)
                wp = torch.zeros(w.shape[0], nfp + i * self.gc, 3, 3)
                wp[:, :self.nf] = w[:, :self.nf]
                wp[:, nfp:] = w[:, self.nf:]
                cs.append(_ConvP(ops, wp, b, f16=self.f16, x3=None if self.f16 else True))
            self._h2 = (nfp, cs)
        return self._h2

    def run_h2(self, ops, ws, hb, tag, x, out):
        """The same block on h2 tensors and the LDS-DMA kernels (round 5: the full-resolution convs of the learned priors): x fp32 [B,nf,H,W] ->
        out (h2 view or fp32 NCHW view).  The growing concat is one h2 buffer of nfp + 4*gc channels."""
        B, _, H, W = x.shape
        nfp, cs = self.h2_ready(ops)
        gc, o = self.gc, (lambda c: c // 8)
        D = hb(tag + "_dense_h2", "h2", B, nfp + 4 * gc, H, W)
        if os.environ.get("BFSR_PRIOR_GLUE", "fused") != "launches" and hasattr(ops, "h2_pack_pad"):
            ops.h2_pack_pad(x, D[:, :o(nfp)])                      # round 6: pad + pack in one launch (the same bits as the two below)
        else:
            xp = hb(tag + "_xpad", "f32z", B, nfp, H, W)           # zero-initialised once: the pad channels stay zero
            ops.axpb_clamp(x, xp[:, :self.nf])
            ops.h2_pack(xp, D[:, :o(nfp)])
        for i in range(4):
            cs[i].run_h2(ops, D[:, :o(nfp + i * gc)], D[:, o(nfp + i * gc): o(nfp + (i + 1) * gc)], act=ACT_LRELU, slope=0.2)
        cs[4].run_h2(ops, D, out)
        return out

    def run(self, ops, ws, tag, x, out):
        B, _, H, W = x.shape
        nf, gc = self.nf, self.gc
        D = ws.get(tag + "_dense", B, nf + 4 * gc, H, W)
        ops.axpb_clamp(x, D[:, :nf])
        for i in range(4):
            self.convs[i].run(ops, D[:, :nf + i * gc], D[:, nf + i * gc: nf + (i + 1) * gc], act=ACT_LRELU, slope=0.2)
        self.convs[4].run(ops, D, out)
        return out


class UNetBody(object):
    """inc -> downs -> ups -> outc, parameter names with a branch suffix `tag` ('' for LINF, '0'/'1' SRFlow)."""

    def __init__(self, ops, sd, tag, depth, f16=False):
        self.ops, self.depth, self.tag = ops, depth, tag
        self.inc = _DoubleConv(ops, sd, "inc%s" % tag, f16=f16)
        self.downs = [_DoubleConv(ops, sd, "down_layers%s.%d.maxpool_conv.1" % (tag, i), f16=f16) for i in range(depth)]
        self.ups = [_DoubleConv(ops, sd, "up_layers%s.%d.conv" % (tag, i), f16=f16) for i in range(depth)]
        self.outc = _ConvP(ops, sd["outc%s.conv.weight" % tag], sd["outc%s.conv.bias" % tag])

    def run(self, ws, x, out, name, top_h2=None):
        """x [B,dim,H,W] -> out [B,Cout,H,W].  top_h2 = (x_h2, hb): the convs of the FULL-resolution level (inc, the last up layer) and of every
        lower level that still has >= 32 tiles of 16 x 32 per sample run on the LDS-DMA kernels over h2 tensors (x_h2 = the input as an h2 view,
        hb = the engine's buffer factory; pooling and bilinear up-sampling stay fp32 kernels: pack / unpack glue around them); the small levels
        stay on the register-staged kernels.  The choice depends on the sample size only, never on the batch."""
        ops, depth = self.ops, self.depth
        B, _, H, W = (x.shape if top_h2 is None else (top_h2[0].shape[0], None, top_h2[0].shape[3], top_h2[0].shape[4]))
        sizes = [(H, W)]
        for i in range(depth):
            sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
        if min(sizes[-1]) < 1:
            raise ValueError("input %dx%d too small for a depth-%d UNet" % (H, W, depth))
        hb = top_h2[1] if top_h2 is not None else None
        lower = os.environ.get("BFSR_PRIOR_LEVELS", "h2") == "h2"
        mt = _min_tiles()
        on_h2 = [top_h2 is not None and (i == 0 or (lower and ((sizes[i][0] + 15) // 16) * ((sizes[i][1] + 31) // 32) >= mt)) for i in range(depth)]
        # skip feature i (i < depth) lives in the first channels of the concat buffer of up layer depth-1-i
        feats, cat_h2 = [], {}
        chans = [self.inc.out] + [d.out for d in self.downs]
        for i in range(depth):
            up = self.ups[depth - 1 - i]
            cat = ws.get("%s_cat%d" % (name, i), B, up.c1.pw.Cin, sizes[i][0], sizes[i][1])
            feats.append(cat)
            if on_h2[i]:
                cat_h2[i] = hb("%s_cat%d_h2" % (name, i), "h2", B, up.c1.pw.Cin, sizes[i][0], sizes[i][1])
        bottom = ws.get("%s_bottom" % name, B, chans[depth], sizes[depth][0], sizes[depth][1])
        # round 6: pooling and up-sampling of the h2 levels as h2 kernels (maxpool2_h2, resize_h2: the same bits as unpack -> pool -> pack and resize -> pack, without
        # the fp32 round trips); BFSR_PRIOR_GLUE=launches keeps those launches (A/B, tests)
        fused = os.environ.get("BFSR_PRIOR_GLUE", "fused") != "launches" and hasattr(ops, "maxpool2_h2")
        if on_h2[0]:
            self.inc.run_h2(ops, hb, "%s_inc" % name, top_h2[0], cat_h2[0][:, :chans[0] // 8], lo=True)
            if not fused:
                ops.h2_unpack(cat_h2[0][:, :chans[0] // 8], feats[0][:, :chans[0]])      # the pooling below reads fp32
        else:
            self.inc.run(ops, ws, "%s_inc" % name, x, feats[0][:, :chans[0]])
        cur = feats[0][:, :chans[0]]
        for i in range(depth):
            pooled = ws.get("%s_pool%d" % (name, i), B, chans[i], sizes[i + 1][0], sizes[i + 1][1])
            nxt_h2 = i + 1 < depth and on_h2[i + 1]
            ph = hb("%s_pool%d_h2" % (name, i), "h2", B, chans[i], sizes[i + 1][0], sizes[i + 1][1]) if nxt_h2 else None
            if fused and on_h2[i]:
                ops.maxpool2_h2(cat_h2[i][:, :chans[i] // 8], out_h2=ph, out_f32=None if nxt_h2 else pooled)
            else:
                ops.maxpool2(cur, pooled)
                if nxt_h2:
                    ops.h2_pack(pooled, ph)
            dst = feats[i + 1][:, :chans[i + 1]] if i + 1 < depth else bottom
            if nxt_h2:
                sk = cat_h2[i + 1][:, :chans[i + 1] // 8]
                self.downs[i].run_h2(ops, hb, "%s_down%d" % (name, i), ph, sk, lo=True)
                if not fused:
                    ops.h2_unpack(sk, dst)
            else:
                self.downs[i].run(ops, ws, "%s_down%d" % (name, i), pooled, dst)
            cur = dst
        for j in range(depth):
            i = depth - 1 - j            # skip level
            cat = feats[i]
            cs = chans[i]
            h1, w1 = cur.shape[2], cur.shape[3]
            Hs, Ws = sizes[i]
            uh, uw = 2 * h1, 2 * w1
            dy, dx = Hs - uh, Ws - uw
            r_h = float(h1 - 1) / float(uh - 1) if uh > 1 else 0.0
            r_w = float(w1 - 1) / float(uw - 1) if uw > 1 else 0.0
            o = ws.get("%s_up%d" % (name, j), B, self.ups[j].out, Hs, Ws)
            if on_h2[i] and fused:
                ops.resize_h2(cur, cat_h2[i][:, cs // 8:], MODE_BILINEAR_AC, r_h, r_w, window=(dy // 2, dx // 2, uh, uw))      # the upsampled half joins the skip half (already h2)
                self.ups[j].run_h2(ops, hb, "%s_upc%d" % (name, j), cat_h2[i], o)
                cur = o
                continue
            ops.resize(cur, cat[:, cs:], MODE_BILINEAR_AC, r_h, r_w, window=(dy // 2, dx // 2, uh, uw))
            if on_h2[i]:
                ops.h2_pack(cat[:, cs:], cat_h2[i][:, cs // 8:])                  # the upsampled half joins the skip half (already h2)
                self.ups[j].run_h2(ops, hb, "%s_upc%d" % (name, j), cat_h2[i], o)
            else:
                self.ups[j].run(ops, ws, "%s_upc%d" % (name, j), cat, o)
            cur = o
        self.outc.run(ops, cur, out)
        return out


class H2Buffers(object):
    """Named h2 / zero-initialised fp32 buffers of the h2 paths (h2 tensors are not fp32: kept outside _Workspace); reallocated on a shape change."""

    def __init__(self, ops):
        self.ops, self.bufs = ops, {}

    def __call__(self, name, kind, *shape):
        t = self.bufs.get(name)
        want = (kind,) + tuple(shape)
        if t is None or t[0] != want:
            t = self.bufs[name] = (want, self.ops.h2_empty(*shape) if kind == "h2" else self.ops.zeros(*shape))
        return t[1]


class SRFlowPriorEngine(object):
    """SRFlow-LP prior `UNet.forward(epses) -> [z0, z1]` (models/unet.py:154-181)."""

    def __init__(self, sd, ops, depth=3):
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if v.dtype.is_floating_point}
        self.ops, self.ws, self._hb = ops, _Workspace(ops), H2Buffers(ops)
        self.proj = [DenseBlock(ops, sd, "input_proj0"), DenseBlock(ops, sd, "input_proj1")]
        self.body = [UNetBody(ops, sd, "0", depth), UNetBody(ops, sd, "1", depth)]

    def _use_h2(self, e):
        """The full-resolution convolutions of a branch (DenseBlock_5C projection, `inc`, the last up layer: 9 of its convs and ~3/4 of its
        arithmetic) on conv_h2x instead of the register-staged split conv (0.45 against 0.30 of the split's matrix-pipe bound) -- when the
        fp16-pair split is active and the latent has at least 32 tiles of 16 x 32 per sample (branch 0: 6 channels at half the HR resolution)."""
        B, C, H, W = e.shape
        return (h2_mode(self.ops, False) == "h2x" and os.environ.get("BFSR_PRIOR", "h2x") == "h2x"
                and ((H + 15) // 16) * ((W + 31) // 32) >= _min_tiles())          # per SAMPLE: the kernel choice must not depend on the batch (bit-identical shards)

    def forward_branch(self, b, e, out=None):
        """Branch b of the prior on latent b (the two branches share nothing, models/unet.py:154-181); `out` may be preallocated by the caller
        (e.g. on another stream than the one this call is enqueued on)."""
        B, C, H, W = e.shape
        if out is None:
            out = self.ops.empty(B, self.body[b].outc.pw.Cout, H, W)
        if self._use_h2(e):
            p_h2 = self._hb("proj%d_h2" % b, "h2", B, self.proj[b].out, H, W)
            self.proj[b].run_h2(self.ops, self.ws, self._hb, "proj%d" % b, e, p_h2)
            self.body[b].run(self.ws, None, out, "u%d" % b, top_h2=(p_h2, self._hb))
            return out
        p = self.ws.get("proj%d" % b, B, self.proj[b].out, H, W)
        self.proj[b].run(self.ops, self.ws, "proj%d" % b, e, p)
        self.body[b].run(self.ws, p, out, "u%d" % b)
        return out

    def forward(self, epses):
        return [self.forward_branch(b, epses[b]) for b in (0, 1)]
