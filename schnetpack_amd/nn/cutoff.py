"""Mirror of ``schnetpack.nn.cutoff.CosineCutoff`` (nn/cutoff.py:14-57)."""
import math

import torch
from torch import nn

from .. import _lib, ops

__all__ = ["CosineCutoff", "cosine_cutoff"]


def cosine_cutoff(input: torch.Tensor, cutoff: torch.Tensor):
    """0.5 (cos(pi d / rc) + 1) [d < rc] -- differentiable torch formula (training path)."""
    input_cut = 0.5 * (torch.cos(input * math.pi / cutoff) + 1.0)
    return input_cut * (input < cutoff).float()


class CosineCutoff(nn.Module):
    r"""Behler-style cosine cutoff; buffer ``cutoff`` of shape [1] like the reference."""

    def __init__(self, cutoff: float):
        super().__init__()
        self.register_buffer("cutoff", torch.FloatTensor([cutoff]))
        self._cutoff_host = float(cutoff)

    def cutoff_value(self) -> float:
        """Host copy of the cutoff radius (no device sync per call)."""
        if self.__dict__.get("_cutoff_host") is None:
            self.__dict__["_cutoff_host"] = float(self.cutoff.item())
        return self.__dict__["_cutoff_host"]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._cutoff_host = None

    def forward(self, input: torch.Tensor):
        ops._check_float(input, "CosineCutoff")
        if self.training and torch.is_grad_enabled() and input.requires_grad:
            return cosine_cutoff(input, self.cutoff)
        dummy = self.cutoff  # any fp32 device tensor: the kernel ignores p0/p1 when phi is not requested
        return ops.RadialCutoffFn.apply(input, _lib.SPK_RBF_BESSEL, dummy, None, self.cutoff_value(), False, True)
