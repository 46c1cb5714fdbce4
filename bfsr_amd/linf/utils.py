"""`make_coord` (reference LINF-LP/utils.py:105-120): coordinates of the pixel centres of a grid in [-1, 1] (or in
`ranges`).  The per-axis values are produced with the reference's float arithmetic -- python-double `v0 + r`, float32
`(2r) * i` -- because nearest-cell lookups in the query kernels depend on their exact bits."""
import torch


def _axis_centres(n, lo=-1, hi=1):
    half_step = (hi - lo) / (2 * n)
    return lo + half_step + (2 * half_step) * torch.arange(n).float()


def make_coord(shape, ranges=None, flatten=True):
    axes = [_axis_centres(n, *((-1, 1) if ranges is None else ranges[i])) for i, n in enumerate(shape)]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)
    return grid.view(-1, grid.shape[-1]) if flatten else grid
