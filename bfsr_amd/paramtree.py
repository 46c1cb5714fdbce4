"""Build an nn.Module tree whose state_dict() keys/shapes equal a schema (name -> (shape, kind)).

The reference's checkpoint formats are part of the drop-in boundary: models here do
not mirror the reference's class hierarchy, they only expose identically named parameters/buffers so that
`load_state_dict(strict=True)` and `state_dict()` interoperate with reference checkpoints.
"""
import torch
from torch import nn

_BUFFER_KINDS = ("bn_mean", "bn_var", "bn_count")


class ParamNode(nn.Module):
    """Pure container; never called."""


def attach(root, schema, init):
    """Register every schema entry under `root`, creating intermediate ParamNode containers.
    `init(name, shape, kind) -> tensor` provides the initial value."""
    for name, (shape, kind) in schema.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            child = node._modules.get(p)
            if child is None:
                child = ParamNode()
                node.add_module(p, child)
            node = child
        value = init(name, shape, kind)
        if kind in _BUFFER_KINDS:
            node.register_buffer(parts[-1], value)
        else:
            node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))
    return root


def default_init(seed=0):
    """Reference-style defaults: Conv2dZeros/ActNorm zero-initialised (flow.py:80-81, FlowActNorms.py:35-36),
    everything else from the synthetic recipe."""
    import numpy as np
    from . import synth
    rng = np.random.Generator(np.random.PCG64(seed))

    def init(name, shape, kind):
        if kind in ("zeros_w", "zeros_b_affine", "zeros_b_split", "zeros_logs", "an_bias", "an_logs"):
            return torch.zeros(tuple(shape), dtype=torch.float32)
        if kind == "bn_count":
            return torch.tensor(0, dtype=torch.long)
        return synth._fill(rng, shape, kind)

    return init
