"""Mirror of ``schnetpack.nn.utils.replicate_module`` (nn/utils.py:11-18)."""
from typing import Callable

from torch import nn

__all__ = ["replicate_module"]


def replicate_module(module_factory: Callable[[], nn.Module], n: int, share_params: bool):
    if share_params:
        return nn.ModuleList([module_factory()] * n)
    return nn.ModuleList([module_factory() for _ in range(n)])
