"""Mirror of ``schnetpack.nn.blocks.build_mlp`` (nn/blocks.py:12-76)."""
from typing import Callable, Optional, Sequence, Union

import torch
import torch.nn.functional as F
from torch import nn

from .base import Dense

__all__ = ["build_mlp"]


def build_mlp(n_in: int, n_out: int, n_hidden: Optional[Union[int, Sequence[int]]] = None,
              n_layers: int = 2, activation: Callable = F.silu, last_bias: bool = True,
              last_zero_init: bool = False) -> nn.Module:
    if n_hidden is None:
        c = n_in
        sizes = []
        for _ in range(n_layers):
            sizes.append(c)
            c = max(n_out, c // 2)
        sizes.append(n_out)
    else:
        hidden = [n_hidden] * (n_layers - 1) if type(n_hidden) is int else list(n_hidden)
        sizes = [n_in] + hidden + [n_out]
    layers = [Dense(sizes[i], sizes[i + 1], activation=activation) for i in range(n_layers - 1)]
    if last_zero_init:
        layers.append(Dense(sizes[-2], sizes[-1], activation=None, weight_init=torch.nn.init.zeros_, bias=last_bias))
    else:
        layers.append(Dense(sizes[-2], sizes[-1], activation=None, bias=last_bias))
    return nn.Sequential(*layers)
