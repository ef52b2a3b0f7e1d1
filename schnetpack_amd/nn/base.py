"""Mirror of ``schnetpack.nn.base.Dense`` (nn/base.py:14-55) on the fp32-MFMA dense kernel."""
from typing import Callable, Union

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import xavier_uniform_, zeros_

from .. import _lib
from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from .activations import shifted_softplus
from .fallback import note_fallback, use_aten

__all__ = ["Dense", "activation_id"]


def activation_id(activation):
    """Map an activation callable to the kernel epilogue id (None if it is not fusable)."""
    if activation is None or isinstance(activation, nn.Identity):
        return _lib.SPK_ACT_NONE
    if activation is shifted_softplus or getattr(activation, "__name__", "") == "shifted_softplus":
        return _lib.SPK_ACT_SSP
    if activation is F.silu or isinstance(activation, nn.SiLU) or getattr(activation, "__name__", "") == "silu":
        return _lib.SPK_ACT_SILU
    return None


class Dense(nn.Linear):
    r"""y = activation(x W^T + b); same constructor, parameters (``weight``, ``bias``) and
    initialisation (xavier_uniform / zeros) as the reference.

    ``forward`` is one ``torch.ops.spk_hip.dense`` call (TorchScript-able).  Its autograd node serves both
    regimes: a plain first-order backward (eval-mode ``Forces``) runs the input gradient on the HIP kernel and
    forms weight gradients only if that pass asks for them; a recorded backward (``create_graph=True``, training on
    forces) is differentiable torch algebra, exact to second order."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True,
                 activation: Union[Callable, nn.Module] = None,
                 weight_init: Callable = xavier_uniform_, bias_init: Callable = zeros_):
        self.weight_init = weight_init
        self.bias_init = bias_init
        super().__init__(in_features, out_features, bias)
        self.activation = activation
        if self.activation is None:
            self.activation = nn.Identity()
        self._act_id = _act_code(self.activation)

    def __setstate__(self, state):
        # instances restored from reference pickles never ran this __init__
        super().__setstate__(state)
        if "_act_id" not in self.__dict__:
            self._act_id = _act_code(self.activation)

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)

    def forward(self, input: torch.Tensor):
        if use_aten(input):      # host / non-float32 tensors: F.linear + activation, as the reference (nn/base.py:52-55)
            note_fallback()
            return self.activation(F.linear(input, self.weight, self.bias))
        if self._act_id >= 0:
            return torch.ops.spk_hip.dense(input, self.weight, self.bias, self._act_id)
        # unknown activation callable: linear part on the HIP kernel, activation by the caller's function
        return self.activation(torch.ops.spk_hip.dense(input, self.weight, self.bias, 0))


def _act_code(activation) -> int:
    a = activation_id(activation)
    return -1 if a is None else int(a)
