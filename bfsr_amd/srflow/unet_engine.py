"""Learned-prior UNets on the HIP kernels.

SRFlow-LP prior (SRFlow-LP/code/models/unet.py:109-181): two independent branches
  DenseBlock_5C(in -> 64) -> DoubleConv -> 3 x [MaxPool2, DoubleConv] -> 3 x [bilinear x2 (align_corners=True),
  pad, cat([skip, up]), DoubleConv(mid = in/2)] -> 1x1 conv.
BatchNorm runs in eval mode and is folded into the conv epilogue as (v - mean) * (gamma/sqrt(var+eps)) + beta.
The `cat([skip, up])` buffers are allocated up front: the encoder writes its skip features straight into
the first half, the upsampler into the second half.
"""
import torch

from ..ops import ACT_LRELU, ACT_NONE, MODE_BILINEAR_AC
from .engine import _ConvP, _Workspace


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


class DenseBlock(object):
    """DenseBlock_5C (unet.py:10-36): five 3x3 convs over a growing concat, no residual."""

    def __init__(self, ops, sd, p, f16=False):
        self.convs = [_ConvP(ops, sd["%s.conv%d.weight" % (p, i)], sd["%s.conv%d.bias" % (p, i)], f16=f16) for i in range(1, 6)]
        self.nf = self.convs[0].pw.Cin
        self.gc = self.convs[0].pw.Cout
        self.out = self.convs[4].pw.Cout

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

    def run(self, ws, x, out, name):
        """x [B,dim,H,W] -> out [B,Cout,H,W]."""
        ops, depth = self.ops, self.depth
        B, _, H, W = x.shape
        sizes = [(H, W)]
        for i in range(depth):
            sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
        if min(sizes[-1]) < 1:
            raise ValueError("input %dx%d too small for a depth-%d UNet" % (H, W, depth))
        # skip feature i (i < depth) lives in the first channels of the concat buffer of up layer depth-1-i
        feats = []
        chans = [self.inc.out] + [d.out for d in self.downs]
        for i in range(depth):
            up = self.ups[depth - 1 - i]
            cat = ws.get("%s_cat%d" % (name, i), B, up.c1.pw.Cin, sizes[i][0], sizes[i][1])
            feats.append(cat)
        bottom = ws.get("%s_bottom" % name, B, chans[depth], sizes[depth][0], sizes[depth][1])
        self.inc.run(ops, ws, "%s_inc" % name, x, feats[0][:, :chans[0]])
        cur = feats[0][:, :chans[0]]
        for i in range(depth):
            pooled = ws.get("%s_pool%d" % (name, i), B, chans[i], sizes[i + 1][0], sizes[i + 1][1])
            ops.maxpool2(cur, pooled)
            dst = feats[i + 1][:, :chans[i + 1]] if i + 1 < depth else bottom
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
            ops.resize(cur, cat[:, cs:], MODE_BILINEAR_AC, r_h, r_w, window=(dy // 2, dx // 2, uh, uw))
            o = ws.get("%s_up%d" % (name, j), B, self.ups[j].out, Hs, Ws)
            self.ups[j].run(ops, ws, "%s_upc%d" % (name, j), cat, o)
            cur = o
        self.outc.run(ops, cur, out)
        return out


class SRFlowPriorEngine(object):
    """SRFlow-LP prior `UNet.forward(epses) -> [z0, z1]` (models/unet.py:154-181)."""

    def __init__(self, sd, ops, depth=3):
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if v.dtype.is_floating_point}
        self.ops, self.ws = ops, _Workspace(ops)
        self.proj = [DenseBlock(ops, sd, "input_proj0"), DenseBlock(ops, sd, "input_proj1")]
        self.body = [UNetBody(ops, sd, "0", depth), UNetBody(ops, sd, "1", depth)]

    def forward_branch(self, b, e, out=None):
        """Branch b of the prior on latent b (the two branches share nothing, models/unet.py:154-181); `out` may be preallocated by the caller
        (e.g. on another stream than the one this call is enqueued on)."""
        B, C, H, W = e.shape
        p = self.ws.get("proj%d" % b, B, self.proj[b].out, H, W)
        self.proj[b].run(self.ops, self.ws, "proj%d" % b, e, p)
        if out is None:
            out = self.ops.empty(B, self.body[b].outc.pw.Cout, H, W)
        self.body[b].run(self.ws, p, out, "u%d" % b)
        return out

    def forward(self, epses):
        return [self.forward_branch(b, epses[b]) for b in (0, 1)]
