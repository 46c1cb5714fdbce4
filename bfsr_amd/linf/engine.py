"""Host-side schedule of LINF-LP on the HIP kernels (result-preserving restructuring of the reference):

  * the encoder runs once per LR batch (the reference calls `gen_feat` again for decode, LINF-LP/test.py:22,38);
  * `coef` and `freq` (3x3, 64->256 each, linf.py:250-251) are ONE conv 64->512, computed once instead of per
    256-row chunk and per direction;
  * the per-point conditioning (Fourier features -> shared 1x1 MLP -> `affine_info`, linf.py:325-391) is
    computed once and shared by `query_log_p` and `query_rgb` (cache keyed on the feat/coord/cell tensors);
  * `NaiveLinear` inverses are precomputed in fp64 (the reference calls `torch.linalg.solve` per query,
    flow.py:120); log-determinants are not computed (discarded by the LP harness, test.py:43).
  * 256-row query chunking (test.py:26-32) is result-preserving and replaced by the kernel grid.
"""
import os

import torch

from ..ops import ACT_LRELU, ACT_NONE, ACT_RELU, MODE_BILINEAR
from ..srflow.engine import RRDBEncoder, _ConvP, _Workspace, h2_mode
from ..srflow.unet_engine import DenseBlock, H2Buffers, UNetBody


class EDSREncoder(object):
    """EDSR with no_upsampling (LINF-LP/models/edsr.py:134-146): head conv, n ResBlocks (conv-ReLU-conv, *res_scale + x),
    body conv, + head output."""

    def __init__(self, ops, sd, prefix, n_resblocks=16, res_scale=1.0, f16=False):
        self.ops, self.n, self.res_scale = ops, n_resblocks, float(res_scale)
        g = lambda n: sd[prefix + n]
        self.head = _ConvP(ops, g("head.0.weight"), g("head.0.bias"), f16=f16)
        self.blocks = [(_ConvP(ops, g("body.%d.body.0.weight" % i), g("body.%d.body.0.bias" % i), f16=f16),
                        _ConvP(ops, g("body.%d.body.2.weight" % i), g("body.%d.body.2.bias" % i), f16=f16)) for i in range(n_resblocks)]
        self.tail = _ConvP(ops, g("body.%d.weight" % n_resblocks), g("body.%d.bias" % n_resblocks), f16=f16)
        self.nf = self.head.pw.Cout
        self.ws = _Workspace(ops)

    def forward(self, x, out, on_block=None):
        ops = self.ops
        B, _, h, w = x.shape
        x0 = self.ws.get("head", B, self.nf, h, w)
        a, b, t = (self.ws.get(n, B, self.nf, h, w) for n in ("a", "b", "t"))
        self.head.run(ops, x, x0)
        cur = x0
        for i, (c1, c2) in enumerate(self.blocks):
            c1.run(ops, cur, t, act=ACT_RELU)
            nxt = a if cur is not a else b
            c2.run(ops, t, nxt, res1=cur, alpha1=self.res_scale)
            cur = nxt
        self.tail.run(ops, cur, out, res1=x0, alpha1=1.0)
        return out


def make_encoder(ops, sd, encoder_spec, f16=False):
    name, args = encoder_spec["name"], dict(encoder_spec.get("args") or {})
    if name == "rrdb":
        return RRDBEncoder(ops, sd, "encoder.", args.get("nb", 23), nf=args.get("nf", 64), gc=args.get("gc", 32),
                           skip_from_first=True, f16=f16), args.get("nf", 64)
    if name == "edsr-baseline":
        return EDSREncoder(ops, sd, "encoder.", args.get("n_resblocks", 16), args.get("res_scale", 1), f16=f16), args.get("n_feats", 64)
    raise NotImplementedError("encoder '%s' is outside the hot-path scope" % name)


class LINFEngine(object):
    def __init__(self, sd, ops, encoder_spec, flow_layers=10, num_layer=3, hidden_dim=256, patch_size=3, precision="fp32"):
        """precision: 'fp32' (parity path) or 'fp16' = BASELINE config 5: the conv contractions of the encoder, the
        coef/freq conv and the shared MLP run on the fp16 MFMA (fp32 accumulation, fp32 tensors); the flow stays fp32."""
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        f16 = precision == "fp16"
        self.precision = precision
        self.ops, self.ws, self._hb = ops, _Workspace(ops), H2Buffers(ops)
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        self.hidden, self.ps, self.L = hidden_dim, patch_size, flow_layers
        self.D = 3 * patch_size * patch_size
        self.encoder, self.nf = make_encoder(ops, sd, encoder_spec, f16=f16)
        # coef | freq as one conv (both read the same feat)
        self.cf = _ConvP(ops, torch.cat([sd["coef.weight"], sd["freq.weight"]], 0),
                         torch.cat([sd["coef.bias"], sd["freq.bias"]], 0), mtile=2, f16=f16)
        if hasattr(self.encoder, "out_gain") and hasattr(ops, "channel_gain"):
            self.encoder.out_gain = ops.channel_gain(torch.cat([sd["coef.weight"], sd["freq.weight"]], 0))    # the conv that contracts the encoder's output (range check)
        self.phase = ops.vec(sd["phase.weight"])
        # Fourier features + shared MLP as ONE kernel (linf_mlp.hip) when the MLP has the reference's shape (3 hidden layers of
        # 256); otherwise, or on the all-native-fp32 backend (BFSR_CONV=f32): features kernel + 1x1 convs
        self.fused_mlp = (hidden_dim == 256 and num_layer == 3 and hasattr(ops, "linf_mlp")
                          and (f16 or getattr(ops, "conv_mode", "f32") == "x3"))
        self.mlp = []
        if self.fused_mlp:
            names_ = ["layers.%d" % (2 * j) for j in range(num_layer + 1)]
            # the fused kernel hands affine_info to the flow kernel in a private quad-major layout (16-byte stores and loads)
            self.mlp_packed = ops.pack_linf_mlp([sd[n + ".weight"] for n in names_], [sd[n + ".bias"] for n in names_], x3=not f16,
                                                quad_layers=(flow_layers, 3 * patch_size * patch_size))
        else:
            for j in range(num_layer + 1):
                self.mlp.append(_ConvP(ops, sd["layers.%d.weight" % (2 * j)], sd["layers.%d.bias" % (2 * j)], mtile=2, f16=f16))
        names = ["imnet.linears.%d" % i for i in range(flow_layers)] + ["imnet.last"]
        W = torch.stack([sd[n + "._weight"] for n in names])
        self.lin_b = ops.vec(torch.stack([sd[n + ".bias"] for n in names]))
        self.lin_w = ops.vec(W)
        Winv = torch.inverse(W.double()).float()
        self.lin_winv = ops.vec(Winv)
        self.lin_winv_t = ops.vec(Winv.transpose(1, 2).contiguous())      # for the VJP of the inverse (query_rgb_vjp)
        self.logdet_const = float(sum(torch.slogdet(w.detach().cpu().float())[1] for w in W))     # NaiveLinear logabsdet, flow.py:66-70
        self._feat_key = self._feat = None
        self._cond_key = self._cond = None
        self.ai_fmt = 1 if self.fused_mlp else 0

    # ------------------------------------------------------------------------------------------------
    def gen_feat(self, inp):
        # caches keep a reference to the keyed tensors (identity + version), so recycled storage can never alias
        key = (inp, inp._version)
        if not (self._feat_key is not None and self._feat_key[0] is inp and self._feat_key[1] == inp._version):
            B, _, h, w = inp.shape
            out = self.ops.empty(B, self.nf, h, w)
            self.encoder.forward(inp, out)
            self._feat_key, self._feat = key, out
        return self._feat

    def affine_info(self, feat, coord, cell):
        key = (feat, feat._version, coord, coord._version, cell, cell._version)
        k = self._cond_key
        if k is not None and k[0] is feat and k[2] is coord and k[4] is cell and (k[1], k[3], k[5]) == (key[1], key[3], key[5]):
            return self._cond
        ops, ws, HD = self.ops, self.ws, self.hidden
        B, _, h, w = feat.shape
        _, qh, qw, _ = coord.shape
        h2cf = self.fused_mlp and self.precision == "fp16" and os.environ.get("BFSR_CF", "h2") == "h2"
        cf = None
        if (h2_mode(ops, self.precision == "fp16") is not None and self.cf.mode in ("f16", "x3") and os.environ.get("BFSR_CF", "h2") != "reg"
                and ((h + 15) // 16) * ((w + 31) // 32) >= 32):
            # round 5: on the LDS-DMA kernel of the contraction mode over an h2 copy of feat (conv_h2s: 2.4x conv_f16's rate)
            fh = self._hb("feat_h2", "h2", B, self.nf, h, w)
            ops.h2_pack(feat, fh)
            if h2cf:
                # the fused MLP gathers coef|freq per query point: from an h2 tensor the 8 channels of a block are one 16-byte word per
                # plane instead of eight 4-byte gathers (profiles/r05_mlp_ablation.txt: the gathers were 6.9 of its 19.5 ms)
                cf = self.cf.run_h2(ops, fh, self._hb("cf_h2", "h2", B, 2 * HD, h, w), lo=True)
            else:
                cf = self.cf.run_h2(ops, fh, ws.get("cf", B, 2 * HD, h, w))
        else:
            cf = self.cf.run(ops, feat, ws.get("cf", B, 2 * HD, h, w))
        if self.precision != "fp16" and hasattr(ops, "check_channels"):
            ops.check_channels(cf)                      # the fused MLP splits coef * cos / sin features into fp16 pairs
        if self.fused_mlp:
            x = ops.linf_mlp(cf, coord, cell, self.phase, self.mlp_packed, ws.get("affine_info", B, self.mlp_packed[2], qh, qw), HD,
                             x3=self.precision != "fp16")
            self._cond_key, self._cond = key, x
            return x
        feats = ws.get("fourier", B, 4 * HD, qh, qw)
        ops.linf_features(cf, coord, cell, self.phase, feats, HD)
        x = feats
        for j, layer in enumerate(self.mlp):
            last = j == len(self.mlp) - 1
            y = ws.get("mlp%d" % (j & 1) if not last else "affine_info", B, layer.pw.Cout, qh, qw)
            layer.run(ops, x, y, act=ACT_NONE if last else ACT_RELU)
            x = y
        self._cond_key, self._cond = key, x
        return x

    def query_log_p(self, feat, coord, cell, gt, with_logp=False):
        """-> z [B,D,qh,qw]; with_logp: (log_p [B*qh*qw], z) = the reference's return pair (linf.py:319-322; per point
        sum of slogdet(W) + sum log(scale) over the layers + the standard-normal log-prob of z, flow.py:44-55)."""
        ai = self.affine_info(feat, coord, cell)
        z = self.ops.empty(*gt.shape)
        if not with_logp:
            return self.ops.linf_flow(gt, ai, z, self.lin_w, self.lin_b, self.L, reverse=False, ai_fmt=self.ai_fmt)
        lp = self.ops.empty(gt.shape[0] * gt.shape[2] * gt.shape[3])
        self.ops.linf_flow(gt, ai, z, self.lin_w, self.lin_b, self.L, reverse=False, log_p=lp, logdet_const=self.logdet_const, ai_fmt=self.ai_fmt)
        return lp, z

    def query_rgb(self, feat, coord, cell, zmap, inp=None, fold=True):
        """patch model: -> folded prediction [B,3,ps*qh,ps*qw] (no skip; the harness adds it, LINF-LP/test.py:169-171).
        pixel-wise model (ps=1): -> [B,3,qh,qw] WITH the bilinear grid_sample skip of `inp` (linf.py:193-194).
        fold=False (patch model, the engine's own harness): -> the inverse flow's output [B,D,qh,qw] as it is, for ops.linf_fold_skip."""
        ops = self.ops
        ai = self.affine_info(feat, coord, cell)
        B, _, qh, qw = zmap.shape
        p = self.ws.get("flow_out", B, self.D, qh, qw)
        ops.linf_flow(zmap, ai, p, self.lin_winv, self.lin_b, self.L, reverse=True, ai_fmt=self.ai_fmt)
        if self.ps == 1:
            return ops.grid_sample_add(inp, coord, p, ops.empty(B, 3, qh, qw))
        if not fold:
            return p
        img = ops.empty(B, 3, self.ps * qh, self.ps * qw)
        return ops.patch_fold(p, img, self.ps)


    def query_rgb_vjp(self, feat, coord, cell, grad_out):
        """Vector-Jacobian product of `query_rgb` w.r.t. `zmap`: grad_out [B,3,H,W] (patch model: H, W <= ps*qh, ps*qw; the
        cropped part gets zero gradient) or [B,3,qh,qw] (pixel-wise) -> grad_zmap [B,D,qh,qw].  The conditioning is independent of
        z, so the inverse flow is affine in z and its backward is a transposed flow (LINF-LP/train.py:143: the image-space loss
        of the latent module back-propagates through query_rgb); fold's adjoint is the zero-padded unfold."""
        ops = self.ops
        ai = self.affine_info(feat, coord, cell)
        B, qh, qw, _ = coord.shape
        if self.ps == 1:
            gp = grad_out
        else:
            gp = ops.patch_unfold(grad_out, self.ws.get("flow_grad", B, self.D, qh, qw), self.ps)
        return ops.linf_flow(gp, ai, ops.empty(B, self.D, qh, qw), self.lin_winv_t, self.lin_b, self.L, reverse=2, ai_fmt=self.ai_fmt)


class LINFPriorEngine(object):
    """LINF-LP prior `UNet.forward(x, lr)` (LINF-LP/models/unet.py:144-167)."""

    def __init__(self, sd, ops, in_chans, depth=3, dim=64, precision="fp32"):
        """precision='fp16': the prior's 3x3 convs contract on the fp16 MFMA like the LINF model's (BASELINE config 5)."""
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        f16 = precision == "fp16"
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if v.dtype.is_floating_point}
        self.ops, self.ws, self.in_chans, self.dim, self.f16, self._hb = ops, _Workspace(ops), in_chans, dim, f16, H2Buffers(ops)
        self.input_proj = DenseBlock(ops, sd, "input_proj", f16=f16)
        self.lr_w = ops.to_device(sd["lr_proj.0.weight"])
        self.lr_b = ops.vec(sd["lr_proj.0.bias"])
        self.lr_dense = DenseBlock(ops, sd, "lr_proj.2", f16=f16)
        self.body = UNetBody(ops, sd, "", depth, f16=f16)

    def forward(self, x, lr):
        ops, ws = self.ops, self.ws
        if self._use_h2(x):
            return self._forward_h2(x, lr)
        B, C, H, W = x.shape
        half = self.dim // 2
        cat = ws.get("cat", B, self.dim, H, W)
        self.input_proj.run(ops, ws, "ip", x, cat[:, :half])
        _, _, h, w = lr.shape
        oh, ow = (h + 2 - 3) // 3 + 1, (w + 2 - 3) // 3 + 1
        e0 = ws.get("lr0", B, self.in_chans, oh, ow)
        ops.conv_direct(lr, self.lr_w, self.lr_b, e0, 3, 1, act=ACT_LRELU, slope=0.2)
        if (oh, ow) == (H, W):
            self.lr_dense.run(ops, ws, "lp", e0, cat[:, half:])
        else:
            e1 = ws.get("lr1", B, half, oh, ow)
            self.lr_dense.run(ops, ws, "lp", e0, e1)
            ops.resize(e1, cat[:, half:], MODE_BILINEAR, float(oh) / H, float(ow) / W)   # size= given => r = in/out
        out = ops.empty(B, self.in_chans, H, W)
        return self.body.run(ws, cat, out, "u")

    def _use_h2(self, x):
        """The full-resolution 3x3 convs (both DenseBlock_5C projections, `inc`, the last up layer) on the LDS-DMA kernel of the contraction
        mode over h2 tensors instead of the register-staged kernels (precision fp16: conv_h2s runs at 2.4x conv_f16's rate, profiles/README.md) --
        from 32 tiles of 16 x 32 per sample on."""
        B, C, H, W = x.shape
        return (h2_mode(self.ops, self.f16) is not None and os.environ.get("BFSR_PRIOR", "h2") != "reg"
                and ((H + 15) // 16) * ((W + 31) // 32) >= 32)          # per SAMPLE: the kernel choice must not depend on the batch

    def _forward_h2(self, x, lr):
        ops, ws, hb = self.ops, self.ws, self._hb
        B, C, H, W = x.shape
        half = self.dim // 2
        cat_h2 = hb("cat_h2", "h2", B, self.dim, H, W)
        self.input_proj.run_h2(ops, ws, hb, "ip", x, cat_h2[:, :half // 8])
        _, _, h, w = lr.shape
        oh, ow = (h + 2 - 3) // 3 + 1, (w + 2 - 3) // 3 + 1
        e0 = ws.get("lr0", B, self.in_chans, oh, ow)
        ops.conv_direct(lr, self.lr_w, self.lr_b, e0, 3, 1, act=ACT_LRELU, slope=0.2)
        if (oh, ow) == (H, W):
            self.lr_dense.run_h2(ops, ws, hb, "lp", e0, cat_h2[:, half // 8:])
        else:
            e1 = ws.get("lr1", B, half, oh, ow)
            if ((oh + 15) // 16) * ((ow + 31) // 32) >= 32:
                self.lr_dense.run_h2(ops, ws, hb, "lp", e0, e1)
            else:
                self.lr_dense.run(ops, ws, "lp", e0, e1)
            if os.environ.get("BFSR_PRIOR_GLUE", "fused") != "launches" and hasattr(ops, "resize_h2"):
                ops.resize_h2(e1, cat_h2[:, half // 8:], MODE_BILINEAR, float(oh) / H, float(ow) / W)      # = the two launches below, without the fp32 image
            else:
                e2 = ws.get("lr2", B, half, H, W)
                ops.resize(e1, e2, MODE_BILINEAR, float(oh) / H, float(ow) / W)
                ops.h2_pack(e2, cat_h2[:, half // 8:])
        out = ops.empty(B, self.in_chans, H, W)
        return self.body.run(ws, None, out, "u", top_h2=(cat_h2, hb))
