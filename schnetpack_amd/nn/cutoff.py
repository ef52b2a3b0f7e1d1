"""Mirror of ``schnetpack.nn.cutoff.CosineCutoff`` (nn/cutoff.py:14-57)."""
import math
from typing import Optional

import torch
from torch import nn

from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from .fallback import note_fallback, use_aten

__all__ = ["CosineCutoff", "cosine_cutoff"]


def cosine_cutoff(input: torch.Tensor, cutoff: torch.Tensor):
    """0.5 (cos(pi d / rc) + 1) [d < rc] -- the torch formula, on the tensor's own device (functional form of the reference)."""
    input_cut = 0.5 * (torch.cos(input * math.pi / cutoff) + 1.0)
    return input_cut * (input < cutoff).float()


class CosineCutoff(nn.Module):
    r"""Behler-style cosine cutoff; buffer ``cutoff`` of shape [1] like the reference."""

    def __init__(self, cutoff: float):
        super().__init__()
        self.register_buffer("cutoff", torch.FloatTensor([cutoff]))
        self._cutoff_host = float(cutoff)

    def __setstate__(self, state):
        super().__setstate__(state)
        if "_cutoff_host" not in self.__dict__:      # reference pickles: read the buffer once (host copy, no sync per call)
            self._cutoff_host = float(self.cutoff.item())

    def cutoff_value(self) -> float:
        """Host copy of the cutoff radius (no device sync per call)."""
        return self._cutoff_host

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if not self.cutoff.is_meta:
            self._cutoff_host = float(self.cutoff.item())

    def forward(self, input: torch.Tensor):
        p1: Optional[torch.Tensor] = None
        if use_aten(input):      # host / non-float32 tensors: the reference's formula (nn/cutoff.py:30-32)
            note_fallback()
            return cosine_cutoff(input, self.cutoff)
        if self.training and torch.is_grad_enabled() and input.requires_grad:
            # differentiable to the third order on the device (force training differentiates twice)
            return torch.ops.spk_hip.radial_d(input, None, 2, self.cutoff, p1, self._cutoff_host, 0)
        # (the kernel ignores p0/p1 when phi is not requested: any fp32 device tensor will do)
        return torch.ops.spk_hip.radial_cutoff(input, 1, self.cutoff, p1, self._cutoff_host, False, True)[1]
