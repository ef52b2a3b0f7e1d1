"""Mirror of ``schnetpack.nn.activations.shifted_softplus`` (nn/activations.py:9-22)."""
import math

import torch
from torch.nn import functional

__all__ = ["shifted_softplus"]


def shifted_softplus(x: torch.Tensor):
    r"""softplus(x) - ln 2.  Used as an *identifier* by Dense / the fused kernels (the HIP
    kernels apply it in their epilogues); calling it directly evaluates the torch formula on the
    tensor's own device."""
    return functional.softplus(x) - math.log(2.0)
