"""`SRFlowModel` -- the thin eval wrapper the reference's test.py drives
(SRFlow-LP/code/models/SRFlow_model.py:31-277, base_model.py:95-124).  Inference-only surface:
`get_encode_z`, `get_sr`, `get_sr_with_z`, `get_z`, `load_network`, `feed_data`/`test`/`get_current_visuals`.

The reference wraps netG in nn.DataParallel (SRFlow_model.py:53) and therefore reaches the net through
`.module`; here `netG.module` is the net itself (one process per GPU; data parallelism is
`bfsr_amd.dist`)."""
from collections import OrderedDict

import torch

from ..options import opt_get
from ... import rng
from . import networks


class _ModuleAlias(object):
    """Gives `netG.module` without DataParallel."""

    def __init__(self, net):
        self.__dict__['_net'] = net

    @property
    def module(self):
        return self._net

    def __getattr__(self, k):
        return getattr(self._net, k)

    def __call__(self, *a, **k):
        return self._net(*a, **k)


class SRFlowModel(object):
    def __init__(self, opt, step=0, ops=None):
        self.opt = opt
        self.is_train = opt['is_train']
        self.heats = opt_get(opt, ['val', 'heats'])
        self.n_sample = opt_get(opt, ['val', 'n_sample'])
        self._net = networks.define_Flow(opt, step, ops=ops)
        self._net.eval()
        self.netG = _ModuleAlias(self._net)
        self.log_dict = OrderedDict()

    def to(self, device):
        self._net.to(device)
        return self

    # ---- checkpoints (base_model.py:112-124): raw state_dict, optional 'module.' prefix, optional submodule
    def load_network(self, load_path, network=None, strict=True, submodule=None):
        net = self._net
        if submodule is not None:
            net = getattr(net, submodule)
        load_net = torch.load(load_path, map_location='cpu') if isinstance(load_path, str) else load_path
        clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in load_net.items())
        net.load_state_dict(clean, strict=strict)
        self._net._engine = None

    # ---- inference API -----------------------------------------------------------------------------
    def feed_data(self, data, need_GT=True):
        self.var_L = data['LQ']
        if need_GT:
            self.real_H = data['GT']

    def test(self):
        self.fake_H = {}
        for heat in self.heats or [0.0]:
            for i in range(self.n_sample or 1):
                z = self.get_z(heat, seed=None, batch_size=self.var_L.shape[0], lr_shape=self.var_L.shape)
                self.fake_H[(heat, i)], _ = self._net(lr=self.var_L, z=z, eps_std=heat, reverse=True)

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out['LQ'] = self.var_L.detach()[0].float().cpu()
        for heat in self.heats or [0.0]:
            for i in range(self.n_sample or 1):
                out[('SR', heat, i)] = self.fake_H[(heat, i)].detach()[0].float().cpu()
        return out

    def get_encode_z(self, lq, gt, epses=None, add_gt_noise=True):
        z, _, _ = self._net(gt=gt, lr=lq, reverse=False, epses=epses, add_gt_noise=add_gt_noise)
        return z

    def get_encode_z_and_nll(self, lq, gt, epses=None, add_gt_noise=True):
        z, nll, _ = self._net(gt=gt, lr=lq, reverse=False, epses=epses, add_gt_noise=add_gt_noise)
        return z, nll

    def get_sr(self, lq, heat=None, seed=None, z=None, epses=None):
        return self.get_sr_with_z(lq, heat, seed, z, epses)[0]

    def get_sr_with_z(self, lq, heat=None, seed=None, z=None, epses=None):
        if z is None and epses is None:
            z = self.get_z(heat, seed, batch_size=lq.shape[0], lr_shape=lq.shape)
        sr, _ = self._net(lr=lq, z=z, eps_std=heat, reverse=True, epses=epses, reverse_with_grad=True)
        return sr, z

    def get_z(self, heat, seed=None, batch_size=1, lr_shape=None):
        """z ~ N(0, heat^2) of the final-latent shape (SRFlow_model.py:224-237, split.enable branch)."""
        if seed:
            torch.manual_seed(seed)
        if not opt_get(self.opt, ['network_G', 'flow', 'split', 'enable']):
            raise NotImplementedError("get_z without flow.split.enable")
        fu = self._net.flowUpsamplerNet
        C = fu.C
        H = int(self.opt['scale'] * lr_shape[2] // fu.scaleH)
        W = int(self.opt['scale'] * lr_shape[3] // fu.scaleW)
        # generated directly on the device (the reference samples on the CPU and copies, SRFlow_model.py:229-231)
        dev = self._net.engine().ops.device
        if heat and heat > 0:
            return rng.randn((batch_size, C, H, W), dev) * heat
        return torch.zeros((batch_size, C, H, W), device=dev)
