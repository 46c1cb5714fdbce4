"""`LINFPatch` ('linf-patch') and `LINF` ('linf') -- drop-ins for LINF-LP/models/linf.py:11-428.

`forward(op, inp, feat, coord, cell, gt, temperature, zmap)` with op in {"gen_feat", "query_log_p", "query_rgb",
"log_p", "rgb"}; nested `encoder_spec` / `imnet_spec` are resolved through the registry like the reference
(linf.py:225,242); state_dict keys are the reference's (`encoder.*`, `coef`, `freq`, `phase`, `layers.{0,2,4,6}`,
`imnet.linears.{i}.{bias,_weight}`, `imnet.last.*`).  `query_log_p` returns the reference's pair (log_p per query point, z);
the LP harness only uses z (LINF-LP/test.py:43)."""
import torch

from ... import rng
from torch import nn

from ... import paramtree
from ...guard import EngineHost, run_guarded
from .. import spec
from ..engine import LINFEngine
from .models import make as _make, register


@register('linf-patch')
class LINFPatch(nn.Module, EngineHost):
    def __init__(self, encoder_spec, imnet_spec=None, flow_layers=10, num_layer=3, hidden_dim=256, patch_size=3, ops=None,
                 precision="fp32"):
        super(LINFPatch, self).__init__()
        self.patch_size = patch_size
        self.encoder = _make(encoder_spec)
        self.imnet = _make(imnet_spec, args={'flow_layers': flow_layers, 'patch_size': patch_size})
        full = spec.linf_schema(self.encoder.spec, flow_layers, num_layer, hidden_dim, patch_size)
        own = type(full)((k, v) for k, v in full.items() if not (k.startswith("encoder.") or k.startswith("imnet.")))
        paramtree.attach(self, own, paramtree.default_init(14))
        # keep the reference's key order: encoder.*, coef, freq, phase, layers.*, imnet.*
        self._modules["imnet"] = self._modules.pop("imnet")
        self._cfg = dict(encoder_spec=self.encoder.spec, flow_layers=flow_layers, num_layer=num_layer,
                         hidden_dim=hidden_dim, patch_size=patch_size, precision=precision)
        self._ops, self._engine, self._fb_engine = ops, None, None

    def load_state_dict(self, state_dict, strict=True):
        r = super(LINFPatch, self).load_state_dict(state_dict, strict=strict)
        self._drop_engines()
        return r

    def _apply(self, fn, *a, **k):
        r = super(LINFPatch, self)._apply(fn, *a, **k)
        self._drop_engines()
        return r

    def _build_engine(self, ops):
        return LINFEngine(self.state_dict(), ops, **self._cfg)

    # ---- reference ops -----------------------------------------------------------------------------
    def gen_feat(self, inp):
        e = self.engine()
        return e.gen_feat(e.ops.to_device(inp))

    def query_log_p(self, inp, feat, coord, cell, gt):
        e = self.engine()
        d = e.ops.to_device
        return e.query_log_p(d(feat), d(coord), d(cell), d(gt), with_logp=True)

    def query_rgb(self, inp, feat, coord, cell, temperature=0, zmap=None):
        e = self.engine()
        d = e.ops.to_device
        coord = d(coord)
        if zmap is None:     # tau path (linf.py:398): z ~ N(0,1) * temperature, sampled on device (plumbing)
            B, qh, qw, _ = coord.shape
            zmap = rng.randn((B, qh * qw, e.D), coord.device).view(B, qh, qw, e.D).permute(0, 3, 1, 2).contiguous() * temperature
        return e.query_rgb(d(feat), coord, d(cell), d(zmap), inp=None if inp is None else d(inp))

    def log_p(self, inp, coord, cell, gt):
        return self.query_log_p(inp, self.gen_feat(inp), coord, cell, gt)

    def rgb(self, inp, coord, cell, temperature=0, zmap=None):
        return self.query_rgb(inp, self.gen_feat(inp), coord, cell, temperature, zmap)

    def forward(self, op, inp=None, feat=None, coord=None, cell=None, gt=None, temperature=0, zmap=None):
        if op == "query_rgb" and zmap is not None and torch.is_grad_enabled() and zmap.requires_grad:
            # latent-module training (LINF-LP/train.py:143): the frozen model's query_rgb is differentiable w.r.t. zmap
            return _QueryRGB.apply(zmap, self, inp, feat, coord, cell)
        if op not in ("query_log_p", "query_rgb", "log_p", "rgb", "gen_feat"):
            raise ValueError("unknown op %r" % (op,))

        def run():
            with torch.no_grad():
                if op == "query_log_p":
                    return self.query_log_p(inp, feat, coord, cell, gt)
                if op == "query_rgb":
                    return self.query_rgb(inp, feat, coord, cell, temperature, zmap)
                if op == "log_p":
                    return self.log_p(inp, coord, cell, gt)
                if op == "rgb":
                    return self.rgb(inp, coord, cell, temperature, zmap)
                return self.gen_feat(inp)
        # range guard of the fp16-pair split with automatic bf16x3 re-run (guard.py); inside lp_infer the outer guard owns the flag
        return run_guarded([self], run)


class _QueryRGB(torch.autograd.Function):
    """`query_rgb` with a backward into `zmap` only (the model is frozen while the latent module trains): forward = the HIP inverse
    flow, backward = its transposed flow (engine.query_rgb_vjp)."""

    @staticmethod
    def forward(ctx, zmap, model, inp, feat, coord, cell):
        e = model.engine()
        d = e.ops.to_device
        feat, coord, cell = d(feat.detach()), d(coord), d(cell)
        ctx.model, ctx.saved = model, (feat, coord, cell)
        return e.query_rgb(feat, coord, cell, d(zmap.detach()), inp=None if inp is None else d(inp.detach()))

    @staticmethod
    def backward(ctx, grad_out):
        e = ctx.model.engine()
        feat, coord, cell = ctx.saved
        gz = e.query_rgb_vjp(feat, coord, cell, e.ops.to_device(grad_out.contiguous()))
        return gz, None, None, None, None, None


@register('linf')
class LINF(LINFPatch):
    """Pixel-wise variant (patch_size 1, linf.py:11-216): D = 3 flow per query pixel, `query_rgb` adds the bilinear
    `grid_sample` skip of `inp` itself (linf.py:193-194) and there is no fold."""

    def __init__(self, encoder_spec, imnet_spec=None, flow_layers=10, num_layer=3, hidden_dim=256, ops=None, precision="fp32"):
        super(LINF, self).__init__(encoder_spec, imnet_spec, flow_layers, num_layer, hidden_dim, patch_size=1, ops=ops,
                                   precision=precision)
