"""Mirror of ``schnetpack.nn.radial`` (nn/radial.py:11-110): GaussianRBF, BesselRBF."""
from math import pi
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib
from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from .fallback import note_fallback, use_aten

__all__ = ["gaussian_rbf", "GaussianRBF", "BesselRBF"]


def gaussian_rbf(inputs: torch.Tensor, offsets: torch.Tensor, widths: torch.Tensor):
    """exp(-0.5 / w^2 (d - mu)^2) -- torch formula on the tensor's own device; differentiable to
    any order (used on the training path and for ``trainable=True``)."""
    coeff = -0.5 / torch.pow(widths, 2)
    diff = inputs[..., None] - offsets
    return torch.exp(coeff * torch.pow(diff, 2))


class GaussianRBF(nn.Module):
    r"""Gaussian radial basis functions; buffers/params ``widths``, ``offsets``, attr ``n_rbf``.
    Eval: HIP kernel with a fused first-order backward; training (force loss => double backward): the HIP operator
    ``radial_d`` whose derivatives of every order are again HIP launches; trainable basis parameters: the torch formula."""

    def __init__(self, n_rbf: int, cutoff: float, start: float = 0.0, trainable: bool = False):
        super().__init__()
        self.n_rbf = n_rbf
        offset = torch.linspace(start, cutoff, n_rbf)
        widths = torch.FloatTensor(torch.abs(offset[1] - offset[0]) * torch.ones_like(offset))
        self.trainable = trainable
        if trainable:
            self.widths = nn.Parameter(widths)
            self.offsets = nn.Parameter(offset)
        else:
            self.register_buffer("widths", widths)
            self.register_buffer("offsets", offset)

    def __setstate__(self, state):
        super().__setstate__(state)
        if "trainable" not in self.__dict__:      # reference pickles carry no such attribute
            self.trainable = isinstance(self.offsets, nn.Parameter)

    def kernel_params(self) -> Tuple[int, torch.Tensor, Optional[torch.Tensor]]:
        """(kind, p0, p1) of ``spk_radial_t``: gaussian = (0, offsets, widths)."""
        return 0, self.offsets, self.widths

    def forward(self, inputs: torch.Tensor):
        # training-mode graphs (any order in d; with ``trainable=True`` also the gradients w.r.t. offsets / widths, nn/radial.py:40-45)
        # run the closed operator family spk_hip::radial_d / radial_c; plain evaluation the fused radial + cutoff kernel
        if use_aten(inputs):      # host / non-float32 tensors: the reference's formula (nn/radial.py:11-14)
            note_fallback()
            return gaussian_rbf(inputs, self.offsets, self.widths)
        if torch.is_grad_enabled() and ((self.training and inputs.requires_grad) or (self.trainable and self.offsets.requires_grad)):
            return torch.ops.spk_hip.radial_d(inputs, None, 0, self.offsets, self.widths, 1.0, 0)
        return torch.ops.spk_hip.radial_cutoff(inputs, 0, self.offsets, self.widths, 1.0, True, False)[0]


class BesselRBF(nn.Module):
    """sin(k pi d / rc) / d  (0th order Bessel); buffer ``freqs``, attr ``n_rbf``."""

    def __init__(self, n_rbf: int, cutoff: float):
        super().__init__()
        self.n_rbf = n_rbf
        freqs = torch.arange(1, n_rbf + 1) * pi / cutoff
        self.register_buffer("freqs", freqs)

    def kernel_params(self) -> Tuple[int, torch.Tensor, Optional[torch.Tensor]]:
        """(kind, p0, p1) of ``spk_radial_t``: bessel = (1, freqs, None)."""
        p1: Optional[torch.Tensor] = None
        return 1, self.freqs, p1

    def forward(self, inputs: torch.Tensor):
        p1: Optional[torch.Tensor] = None
        if use_aten(inputs):      # host / non-float32 tensors: the reference's formula (nn/radial.py:105-109)
            note_fallback()
            a = self.freqs[None, :]
            inputs_ = inputs[..., None]
            ax = inputs_ * a
            sinax = torch.sin(ax)
            norm = torch.where(inputs_ == 0, torch.tensor(1.0, device=inputs_.device, dtype=inputs_.dtype), inputs_)
            return sinax / norm
        if self.training and torch.is_grad_enabled() and inputs.requires_grad:
            return torch.ops.spk_hip.radial_d(inputs, None, 1, self.freqs, p1, 1.0, 0)
        return torch.ops.spk_hip.radial_cutoff(inputs, 1, self.freqs, p1, 1.0, True, False)[0]


assert _lib.SPK_RBF_GAUSSIAN == 0 and _lib.SPK_RBF_BESSEL == 1
