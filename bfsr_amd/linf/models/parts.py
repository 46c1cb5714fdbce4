"""Registry entries for the nested specs of LINF (`encoder_spec`, `imnet_spec`): 'rrdb', 'edsr-baseline', 'flow'
(reference LINF-LP/models/rrdb.py:119-128, edsr.py:168-181, flow.py:10-26).  They are parameter holders with the
reference's parameter names; the computation is scheduled by `bfsr_amd.linf.engine`."""
from torch import nn

from ... import paramtree
from .. import spec
from .models import register


class _Holder(nn.Module):
    def __init__(self, schema, seed):
        super(_Holder, self).__init__()
        paramtree.attach(self, schema, paramtree.default_init(seed))

    def forward(self, *a, **k):
        raise RuntimeError("this module only holds parameters; run it through the LINF / LINFPatch model")


@register('rrdb')
def make_rrdb(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, no_upsampling=True):
    if not no_upsampling:
        raise NotImplementedError("rrdb with upsampling is not on the LINF hot path")
    m = _Holder(spec.rrdb_schema("", in_nc, out_nc, nf, nb, gc), 11)
    m.out_dim = nf
    m.spec = {"name": "rrdb", "args": dict(in_nc=in_nc, out_nc=out_nc, nf=nf, nb=nb, gc=gc, no_upsampling=True)}
    return m


@register('edsr-baseline')
def make_edsr_baseline(n_resblocks=16, n_feats=64, res_scale=1, scale=2, no_upsampling=False, rgb_range=1):
    if not no_upsampling:
        raise NotImplementedError("edsr-baseline with a tail is not on the LINF hot path")
    m = _Holder(spec.edsr_schema("", n_resblocks, n_feats), 12)
    m.out_dim = n_feats
    m.spec = {"name": "edsr-baseline", "args": dict(n_resblocks=n_resblocks, n_feats=n_feats, res_scale=res_scale,
                                                     no_upsampling=True)}
    return m


@register('flow')
class Flow(_Holder):
    def __init__(self, flow_layers=10, patch_size=1, name='flow'):
        super(Flow, self).__init__(spec.flow_schema("", flow_layers, patch_size), 13)
        self.n_layers, self.ps_square = flow_layers, patch_size * patch_size
        self.affine_eps = 0.0001
