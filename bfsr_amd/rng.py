"""Gaussian noise source of the stochastic (temperature / tau) paths: SRFlow `get_z` (SRFlow_model.py:224-237) and Split2d's eps
sampling (Split.py:66-70, flow.py:113-119), LINF `torch.randn(...) * temperature` (linf.py:397-398).

Default: `torch.randn` on the device (Philox) -- the reference draws on the CPU and copies.  `set_source` lets a caller (the
parity tests) supply the standard-normal draws, e.g. noise recorded while the genuine reference ran, so that the sampling
paths can be compared value for value."""
import torch

_source = None


def set_source(fn):
    """fn(shape_tuple, device) -> tensor of that shape, or None to fall through to torch.randn; set_source(None) restores the default."""
    global _source
    _source = fn


def randn(shape, device):
    shape = tuple(int(s) for s in shape)
    if _source is not None:
        t = _source(shape, device)
        if t is not None:
            if tuple(t.shape) != shape:
                raise ValueError("noise source returned shape %s for a request of %s" % (tuple(t.shape), shape))
            return t.to(device=device, dtype=torch.float32)
    return torch.randn(*shape, device=device, dtype=torch.float32)


def snapshot(device):
    """State of the default noise source (the device's Philox generator) -- guard.run_guarded takes it before a guarded pass and restores it before
    a re-run, so that a pass repeated under the bf16x3 split (or on per-conv launches) draws the SAME noise: the returned sample stays reproducible
    from the seed.  An injected source (`set_source`) is the caller's to rewind: None."""
    if _source is not None:
        return None
    dev = torch.device(device)
    if dev.type != "cuda":
        return None
    return (dev, torch.cuda.get_rng_state(dev))


def restore(state):
    if state is not None:
        torch.cuda.set_rng_state(state[1], state[0])
