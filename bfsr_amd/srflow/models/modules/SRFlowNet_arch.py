"""`SRFlowNet` -- drop-in for the reference generator
(SRFlow-LP/code/models/modules/SRFlowNet_arch.py:30-158), found by name through
`networks.find_model_using_name('SRFlowNet')` exactly like the reference (networks.py:27-43).

Constructor and `forward` signatures, the `epses` list convention (encode appends
`[eps_split..., z_final]`, FlowUpsamplerNet.py:247-251; decode copies then pops from the end, :206,268,301)
and the state_dict key names are the reference's.  The computation is scheduled by
`bfsr_amd.srflow.engine.SRFlowEngine` on the HIP kernels.

`forward(reverse=False)` returns `(epses | z, nll, logdet)` and `forward(reverse=True)` `(sr, logdet)` with the reference's
values (SRFlowNet_arch.py:83-116,145-158): the flow's log-determinant is accumulated per sample in float64 on the device
(`bfsr_logscale_sum`, `bfsr_gaussian_logp`) and returned as float32.  The LP harness (`test.py`) calls the engine directly
and skips these reductions, as the reference discards the values there (test.py:139)."""
import numpy as np
import torch
from torch import nn

from .... import paramtree
from ....guard import EngineHost, run_guarded
from ... import spec
from ...engine import SRFlowEngine
from ...options import opt_get


class SRFlowNet(nn.Module, EngineHost):
    def __init__(self, in_nc, out_nc, nf, nb, gc=32, scale=4, K=None, opt=None, step=None, ops=None):
        super(SRFlowNet, self).__init__()
        self.opt = opt
        q = opt_get(opt, ['datasets', 'train', 'quant'])
        self.quant = 255 if q is None else q
        self.nb = nb
        if opt['scale'] != scale:
            raise ValueError("scale argument and opt['scale'] disagree")
        schema = spec.rrdb_schema(opt, nb, nf=nf, gc=gc, in_nc=in_nc, out_nc=out_nc)
        schema.update(spec.flow_schema(opt))
        paramtree.attach(self, schema, paramtree.default_init(0))
        fu = self.flowUpsamplerNet
        fu.C = spec.final_channels(opt)                       # FlowUpsamplerNet.C after the last split
        n_sq = sum(1 for ly in spec.flow_layers(opt) if ly.type == "squeeze")
        fu.scaleH = fu.scaleW = float(1 << n_sq)              # 160 / H_final (FlowUpsamplerNet.py:112-115)
        self.RRDB_training = True
        self._ops, self._engine, self._fb_engine = ops, None, None

    # ---- engine lifecycle -------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        r = super(SRFlowNet, self).load_state_dict(state_dict, strict=strict)
        self._drop_engines()
        return r

    def _apply(self, fn, *a, **k):
        r = super(SRFlowNet, self)._apply(fn, *a, **k)
        self._drop_engines()
        return r

    def _build_engine(self, ops):
        return SRFlowEngine(self.opt, self.state_dict(), ops, nb=self.nb)

    def set_rrdb_training(self, trainable):
        self.RRDB_training = trainable
        return False

    # ---- reference API --------------------------------------------------------------------------
    def forward(self, gt=None, lr=None, z=None, eps_std=None, reverse=False, epses=None, reverse_with_grad=False,
                lr_enc=None, add_gt_noise=False, step=None, y_label=None):
        # range guard of the fp16-pair split with automatic bf16x3 re-run (guard.py); the caller's `epses` list is only extended by the pass
        # whose result is returned, and a sampled z is drawn once (SRFlowModel.get_z) -- a re-run sees the same inputs
        n_eps = len(epses) if isinstance(epses, list) else None

        def run():
            eng = self.engine()
            dev = eng.ops.to_device
            if n_eps is not None:
                del epses[n_eps:]
            with torch.no_grad():
                if not reverse:
                    return self.normal_flow(dev(gt), dev(lr), epses=epses, add_gt_noise=add_gt_noise)
                assert lr.shape[1] == 3
                return self.reverse_flow(dev(lr), z, eps_std=eps_std, epses=epses, add_gt_noise=add_gt_noise)
        return run_guarded([self], run)

    def normal_flow(self, gt, lr, y_onehot=None, epses=None, lr_enc=None, add_gt_noise=True, step=None):
        eng = self.engine()
        z = gt
        if add_gt_noise:    # SRFlowNet_arch.py:93-99 (training-time dequantisation noise; plumbing on torch)
            if opt_get(self.opt, ['network_G', 'flow', 'augmentation', 'noiseQuant'], True):
                z = z + ((torch.rand(z.shape, device=z.device) - 0.5) / self.quant)
        ops = eng.ops
        pixels = int(gt.shape[2] * gt.shape[3])
        acc = ops.zeros_f64(gt.shape[0])
        if add_gt_noise:
            acc += float(-np.log(self.quant) * pixels)
        out = eng.encode(z, lr, logdet=acc)
        objective = ops.gaussian_logp(out[-1], acc.clone())            # + GaussianDiag.logp(None, None, z), :108
        nll = ((-objective) / float(np.log(2.) * pixels)).float()
        logdet = acc.float()
        if isinstance(epses, list):
            epses.extend(out)
            return epses, nll, logdet
        return out[-1], nll, logdet

    def reverse_flow(self, lr, z, y_onehot=None, eps_std=None, epses=None, lr_enc=None, add_gt_noise=True):
        eng = self.engine()
        acc = eng.ops.zeros_f64(lr.shape[0])
        if add_gt_noise:       # SRFlowNet_arch.py:149-150
            acc -= float(-np.log(self.quant) * int(lr.shape[2] * lr.shape[3]) * self.opt['scale'] ** 2)
        if isinstance(epses, list):
            sr = eng.decode(lr, epses=[eng.ops.to_device(e) for e in epses], logdet=acc)
        else:
            sr = eng.decode(lr, z=eng.ops.to_device(z), eps_std=eps_std, logdet=acc)
        return sr, acc.float()
