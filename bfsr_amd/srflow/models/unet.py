"""SRFlow-LP learned prior, registry name 'unet' (reference SRFlow-LP/code/models/unet.py:109-186).

`UNet.forward([e0, e1]) -> [z0, z1]`; parameters are named exactly as in the reference so its checkpoints
(`{'prior_model': {'name','args','sd'}}`, test.py:90-91) load with strict=True.  All arithmetic runs in the
HIP kernels via `bfsr_amd.srflow.unet_engine`."""
import torch
from torch import nn

from ... import paramtree
from ...guard import EngineHost
from .. import spec
from ..unet_engine import SRFlowPriorEngine
from .models import register


class UNet(nn.Module, EngineHost):
    def __init__(self, depth=3, dim=64, bilinear=True, ops=None):
        super(UNet, self).__init__()
        self.depth, self.dim, self.bilinear = depth, dim, bilinear
        paramtree.attach(self, spec.srflow_prior_schema(depth, dim, bilinear), paramtree.default_init(1))
        self._ops, self._engine, self._fb_engine = ops, None, None

    def _invalidate(self, *a, **k):
        self._drop_engines()

    def load_state_dict(self, state_dict, strict=True):
        r = super(UNet, self).load_state_dict(state_dict, strict=strict)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super(UNet, self)._apply(fn, *a, **k)
        self._invalidate()
        return r

    def _build_engine(self, ops):
        return SRFlowPriorEngine(self.state_dict(), ops, self.depth)

    def forward(self, epses):
        if self.training:
            raise NotImplementedError("inference engine: call .eval() (BatchNorm runs on running statistics)")
        with torch.no_grad():
            return self.engine().forward([self._ops_dev(e) for e in epses])

    def _ops_dev(self, t):
        return self.engine().ops.to_device(t)


@register('unet')
def make_unet(depth, dim=64, bilinear=True, ops=None):
    print('UNet: depth={}, dim={}, bilinear={}'.format(depth, dim, bilinear))
    return UNet(depth=depth, dim=dim, bilinear=bilinear, ops=ops)
