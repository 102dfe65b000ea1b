"""Small helpers with the reference's names (modules/utils.py:3-37).  Evaluation-side only."""
import torch


def log_sum_exp(value, dim=None, keepdim=False):
    """log(sum(exp(value))) along `dim` (all elements when dim is None), max-shifted."""
    if dim is None:
        return torch.logsumexp(value.reshape(-1), dim=0)
    return torch.logsumexp(value, dim=dim, keepdim=keepdim)


def generate_grid(zmin, zmax, dz, device, ndim=2):
    """1-D: (k,1) tensor; 2-D: ((k*k,2) tensor, k) -- same return convention as the reference."""
    axis = torch.arange(zmin, zmax, dz)
    if ndim == 1:
        return axis.unsqueeze(1).to(device)
    if ndim == 2:
        k = axis.numel()
        g1, g2 = torch.meshgrid(axis, axis, indexing="ij")
        return torch.stack((g1.reshape(-1), g2.reshape(-1)), dim=-1).to(device), k
    raise ValueError("ndim must be 1 or 2")
