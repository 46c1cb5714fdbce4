"""`make_coord` (reference LINF-LP/utils.py:105-120): pixel-centre coordinates in [-1,1]; same float arithmetic
(`v0 + r + (2r) * arange(n).float()`) so nearest-cell lookups agree bit for bit."""
import torch


def make_coord(shape, ranges=None, flatten=True):
    coord_seqs = []
    for i, n in enumerate(shape):
        v0, v1 = (-1, 1) if ranges is None else ranges[i]
        r = (v1 - v0) / (2 * n)
        coord_seqs.append(v0 + r + (2 * r) * torch.arange(n).float())
    ret = torch.stack(torch.meshgrid(*coord_seqs, indexing="ij"), dim=-1)
    if flatten:
        ret = ret.view(-1, ret.shape[-1])
    return ret
