"""Mirror of ``schnetpack.nn.scatter`` (nn/scatter.py:7-34) on the HIP path."""
import torch

from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from .fallback import note_fallback, use_aten

__all__ = ["scatter_add"]


def scatter_add(x: torch.Tensor, idx_i: torch.Tensor, dim_size: int, dim: int = 0) -> torch.Tensor:
    """Sum over values with the same indices: ``zeros(shape).index_add(dim, idx_i, x)``.

    Same signature, argument meaning and output shape/dtype/device as the reference.  Sorted
    indices (every reference neighbour list, ``idx_m``) take the deterministic segmented-sum
    kernel; unsorted ones use float atomics.  Differentiable to any order (its backward is the HIP
    ``gather``, whose backward is this function).  TorchScript-able.
    """
    if use_aten(x):          # host / non-float32 tensors: the reference's own formula (nn/scatter.py:30-34)
        note_fallback()
        shape = list(x.shape)
        shape[dim] = dim_size
        tmp = torch.zeros(shape, dtype=x.dtype, device=x.device)
        return tmp.index_add(dim, idx_i, x)
    return torch.ops.spk_hip.scatter_add(x, idx_i, dim_size, dim)
