"""The stochastic (temperature / tau) paths against the GENUINE reference with its noise draws recorded
(tests/golden/srflow_sampling.npz, linf_sampling.npz; generator tests/golden/make_golden_extra.py):
  SRFlow: `SRFlowModel.get_sr_with_z` -> `get_z` (SRFlow_model.py:224-237) + Split2d eps sampling (Split.py:66-70, flow.py:113-119)
  LINF:   `query_rgb(zmap=None, temperature)` (linf.py:397-398) through `batched_predict` (LINF-LP/test.py:20-31)
The product draws with `bfsr_amd.rng.randn` (torch.randn on the device); the tests plug the recorded draws into that hook.
CPU variants run the host logic on the test double, `-m gpu` variants the HIP path.  Tolerance: 1e-4 max-abs (fp32)."""
import os

import numpy as np
import pytest
import torch

from bfsr_amd import rng, synth
from bfsr_amd.srflow import options, spec
from cpu_ops import CpuOps

T = torch.from_numpy


class _Recorded(object):
    """hands out the recorded standard-normal tensors in the order the pipeline asks for them; checks the shapes"""

    def __init__(self, *draws):
        self.draws, self.asked = list(draws), []

    def __call__(self, shape, device):
        self.asked.append(shape)
        t = self.draws.pop(0)
        assert tuple(t.shape) == shape, (tuple(t.shape), shape)
        return t


def _srflow_sampling(ops, golden_dir):
    from bfsr_amd.srflow.models import create_model
    g = np.load(os.path.join(golden_dir, "srflow_sampling.npz"))
    opt = options.load(options.DEFAULT_CONF)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    heat = float(g["heat"])
    lr = T(g["lr"])
    # get_z draws z/heat first, Split2d then draws eps/heat at level 1 (decode walks the layers from the top)
    src = _Recorded(T(g["z"]) / heat, T(g["eps"]) / heat)
    rng.set_source(src)
    try:
        sr, z = m.get_sr_with_z(lr, heat=heat)
    finally:
        rng.set_source(None)
    assert src.asked == [(2, 96, 8, 6), (2, 6, 32, 24)] and not src.draws
    assert (z.cpu() - T(g["z"])).abs().max() <= 1e-6
    ref = T(g["sr"])
    err = (sr.cpu() - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    # the explicit-z form of the same call (caller supplies z, Split2d still samples)
    rng.set_source(_Recorded(T(g["eps"]) / heat))
    try:
        sr2 = m.get_sr(lr, heat=heat, z=T(g["z"]))
    finally:
        rng.set_source(None)
    assert (sr2.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def _linf_sampling(ops, golden_dir):
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import batched_predict
    from bfsr_amd.ops import MODE_BILINEAR
    from test_linf_cpu import mspec, weights
    g = np.load(os.path.join(golden_dir, "linf_sampling.npz"))
    sd, _ = weights("edsr-baseline", 2025)
    m = make(mspec("edsr-baseline"), args={"ops": ops}).eval()
    m.load_state_dict(sd)
    lr = T(g["lr"])
    inp = ops.to_device((lr - 0.5) / 0.5)
    noise = T(g["noise"])
    src = _Recorded(noise.view(1, *noise.shape))
    rng.set_source(src)
    try:
        full = batched_predict(m, inp, T(g["coord"]), T(g["cell"]), float(g["temperature"]))
    finally:
        rng.set_source(None)
    assert not src.draws
    H, W = g["pred"].shape[-2:]
    pred = full[..., :H, :W].contiguous()
    skip = ops.resize(inp, ops.empty(1, 3, H, W), MODE_BILINEAR, float(lr.shape[2]) / H, float(lr.shape[3]) / W)
    raw = (pred + skip).cpu()
    ref = T(g["pred_raw"])
    assert (raw - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert (torch.clamp(raw * 0.5 + 0.5, 0, 1) - T(g["pred"])).abs().max().item() <= 1e-4


def test_srflow_sampling_vs_reference_cpu(golden_dir):
    _srflow_sampling(CpuOps(), golden_dir)


def test_linf_sampling_vs_reference_cpu(golden_dir):
    _linf_sampling(CpuOps(), golden_dir)


@pytest.mark.gpu
def test_srflow_sampling_vs_reference_gpu(golden_dir):
    from bfsr_amd.ops import HipOps
    _srflow_sampling(HipOps("cuda:0"), golden_dir)


@pytest.mark.gpu
def test_linf_sampling_vs_reference_gpu(golden_dir):
    from bfsr_amd.ops import HipOps
    _linf_sampling(HipOps("cuda:0"), golden_dir)


def test_default_noise_source_is_device_randn():
    rng.set_source(None)
    a = rng.randn((2, 3), torch.device("cpu"))
    assert a.shape == (2, 3) and a.dtype == torch.float32
    with pytest.raises(ValueError):
        rng.set_source(lambda s, d: torch.zeros(1))
        try:
            rng.randn((2, 3), torch.device("cpu"))
        finally:
            rng.set_source(None)
