"""Seeded synthetic weights and inputs (no network: the reference's checkpoints are not available).

The recipe is the *conditioned* one of SURVEY.md section 8c: with untrained weights the SRFlow
inverse amplifies errors by up to (1/0.88)^96, so the last Conv2dZeros of every coupling net gets a
small weight std and a +4 bias on its "scale" (odd) channels (sigmoid(h+2) ~ 0.998); invconv
weights are random orthogonal (the reference's own init, Permutations.py:29); LINF NaiveLinear
weights are orthogonal x diag(U[0.8,1.25]).

Everything is drawn from numpy's PCG64 stream in schema order, so the build container and the GPU
box obtain identical tensors from the same seed.
"""
import hashlib

import numpy as np
import torch


def _orthogonal(rng, n):
    q, r = np.linalg.qr(rng.standard_normal((n, n)))
    q = q * np.sign(np.diag(r))          # make the factorisation unique
    return q


def _fill(rng, shape, kind):
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if kind in ("kaiming0.1", "kaiming", "kaiming0.3"):
        fan_in = int(np.prod(shape[1:]))
        std = np.sqrt(2.0 / fan_in) * {"kaiming0.1": 0.1, "kaiming0.3": 0.3, "kaiming": 0.6}[kind]
        a = rng.standard_normal(shape) * std
    elif kind == "meanshift_w":                   # EDSR MeanShift (unused by forward, present in the sd)
        a = np.eye(3).reshape(3, 3, 1, 1)
    elif kind == "linf_last":                     # last MLP layer -> affine_info: keep the flow well conditioned
        fan_in = int(np.prod(shape[1:]))
        a = rng.standard_normal(shape) * np.sqrt(1.0 / fan_in) * 0.3
    elif kind == "linf_last_bias":
        a = rng.standard_normal(shape) * 0.05
    elif kind == "bias_small":
        a = rng.standard_normal(shape) * 0.01
    elif kind == "flowconv":                      # flow.Conv2d init, flow.py:54 (std 0.05)
        a = rng.standard_normal(shape) * 0.05
    elif kind in ("an_bias", "an_logs"):
        a = rng.standard_normal(shape) * 0.05
    elif kind == "orthogonal":
        a = _orthogonal(rng, shape[0])
    elif kind == "zeros_w":
        a = rng.standard_normal(shape) * 2e-3
    elif kind == "zeros_b_affine":                # odd channels = "scale" under the cross split
        a = rng.standard_normal(shape) * 0.05
        a[1::2] += 4.0
    elif kind == "zeros_b_split":                 # odd channels = logs of the split prior: keep ~0
        a = rng.standard_normal(shape) * 0.05
    elif kind == "zeros_logs":
        a = rng.standard_normal(shape) * 0.02
    elif kind == "bn_weight":
        a = rng.uniform(0.8, 1.2, shape)
    elif kind == "bn_bias":
        a = rng.standard_normal(shape) * 0.05
    elif kind == "bn_mean":
        a = rng.standard_normal(shape) * 0.1
    elif kind == "bn_var":
        a = rng.uniform(0.5, 1.5, shape)
    elif kind == "bn_count":
        return torch.tensor(1, dtype=torch.long)
    elif kind == "linf_linear":                   # NaiveLinear._weight: Q * diag(U[0.8,1.25])
        a = _orthogonal(rng, shape[0]) * rng.uniform(0.8, 1.25, shape[0])[None, :]
    elif kind == "linf_bias":
        a = rng.standard_normal(shape) * 0.05
    elif kind == "default_conv":                  # torch default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        a = rng.uniform(-b, b, shape)
    elif kind == "default_bias":
        a = rng.uniform(-0.05, 0.05, shape)
    else:
        raise KeyError(kind)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(shape))


def state_dict_from_schema(schema, seed):
    """schema: OrderedDict name -> (shape, kind).  Returns an OrderedDict of CPU fp32 tensors."""
    from collections import OrderedDict
    rng = np.random.Generator(np.random.PCG64(seed))
    return OrderedDict((name, _fill(rng, shape, kind)) for name, (shape, kind) in schema.items())


def digest(sd):
    """sha256 over the raw bytes of a state_dict (weight-identity check across machines)."""
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def lr_batch(seed, B, h, w):
    """Synthetic LR batch in [0,1): Generator(PCG64(seed)).random((B,3,h,w), float32)
    (SURVEY.md section 8d)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random((B, 3, h, w), dtype=np.float32))


def smooth_lr_batch(seed, B, h, w):
    """A smoother LR batch (low-pass of uniform noise, rescaled to [0,1]) -- closer to image
    statistics; used by some parity tests beside the white-noise batch."""
    x = lr_batch(seed, B, h, w)
    k = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0])
    k = (k[:, None] * k[None, :]) / 256.0
    x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"),
                                   k.expand(3, 1, 5, 5).contiguous(), groups=3)
    lo = x.amin(dim=(1, 2, 3), keepdim=True)
    hi = x.amax(dim=(1, 2, 3), keepdim=True)
    return ((x - lo) / (hi - lo + 1e-12)).contiguous()
