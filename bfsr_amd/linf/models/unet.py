"""LINF-LP learned prior, registry name 'unet' (reference LINF-LP/models/unet.py:105-172):
`UNet.forward(z [B,27,Q,Q], lr [B,3,h,w]) -> [B,27,Q,Q]`."""
import torch
from torch import nn

from ... import paramtree
from ...guard import EngineHost
from .. import spec
from ..engine import LINFPriorEngine
from .models import register


class UNet(nn.Module, EngineHost):
    def __init__(self, in_chans, depth=3, dim=64, bilinear=False, ops=None, precision="fp32"):
        super(UNet, self).__init__()
        self.in_chans, self.depth, self.dim, self.bilinear, self.precision = in_chans, depth, dim, bilinear, precision
        paramtree.attach(self, spec.linf_prior_schema(in_chans, depth, dim, bilinear), paramtree.default_init(15))
        self._ops, self._engine, self._fb_engine = ops, None, None

    def load_state_dict(self, state_dict, strict=True):
        r = super(UNet, self).load_state_dict(state_dict, strict=strict)
        self._drop_engines()
        return r

    def _apply(self, fn, *a, **k):
        r = super(UNet, self)._apply(fn, *a, **k)
        self._drop_engines()
        return r

    def _build_engine(self, ops):
        return LINFPriorEngine(self.state_dict(), ops, self.in_chans, self.depth, self.dim, precision=self.precision)

    def forward(self, x, lr):
        if self.training:
            raise NotImplementedError("inference engine: call .eval()")
        e = self.engine()
        with torch.no_grad():
            return e.forward(e.ops.to_device(x), e.ops.to_device(lr))


@register('unet')
def make_unet(in_chans, depth=3, dim=64, bilinear=True, cell_input=None, ops=None, precision="fp32"):
    print('UNet: depth={}, dim={}, bilinear={}'.format(depth, dim, bilinear))
    return UNet(in_chans=in_chans, depth=depth, dim=dim, bilinear=bilinear, ops=ops, precision=precision)
