"""Mirrors of the reference *callers* either side of the hot path, so that a complete force call can be
assembled (and scripted) without the reference package: ``PairwiseDistances`` (atomistic/distances.py:9-26),
``Atomwise`` (atomistic/atomwise.py:14-88) and ``Forces`` (atomistic/response.py:18-92).  In an integration the
reference's own modules run unchanged on top of the HIP classes (tests/test_gpu_reference_callers.py) -- they only see
``schnetpack.nn.scatter_add`` / ``Dense`` / the representation classes; ``install(fused_head=True)`` swaps in this
``Atomwise`` for its one-kernel energy head.
"""
from typing import Callable, Dict, Final, List, Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, properties
from . import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from .nn import Dense, build_mlp, scatter_add
from .nn.base import activation_id
from .nn.fallback import note_fallback, use_aten

__all__ = ["PairwiseDistances", "Atomwise", "Forces"]


class PairwiseDistances(nn.Module):
    """Rij = R[idx_j] - R[idx_i] + offsets; autograd lands dE/dRij back on the atoms (segmented row sum on
    sorted symmetric lists) and on the offsets (stress via ``Strain``)."""

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        R = inputs[properties.R]
        offsets: Optional[torch.Tensor] = None
        if properties.offsets in inputs:
            offsets = inputs[properties.offsets]
        idx_i = inputs[properties.idx_i]
        idx_j = inputs[properties.idx_j]
        if use_aten(R):          # host / non-float32 tensors: the reference's formula (atomistic/distances.py:19-25)
            note_fallback()
            Rij = R[idx_j] - R[idx_i]
            inputs[properties.Rij] = Rij + offsets if offsets is not None else Rij
            return inputs
        inputs[properties.Rij] = torch.ops.spk_hip.pairwise(R, idx_i, idx_j, offsets)
        return inputs


class Atomwise(nn.Module):
    """Per-atom MLP + sum over ``idx_m`` (aggregation_mode 'sum' / 'avg' / None).  In eval mode the default head
    (2 layers, width-1 output) is ONE kernel each way (``torch.ops.spk_hip.atomwise``)."""

    _fused_head: Final[bool]

    def __init__(self, n_in: int, n_out: int = 1, n_hidden: Optional[Union[int, Sequence[int]]] = None,
                 n_layers: int = 2, activation: Callable = F.silu, aggregation_mode: str = "sum",
                 output_key: str = "y", per_atom_output_key: Optional[str] = None,
                 n_molecules_key: str = "_n_molecules"):
        super().__init__()
        self.output_key = output_key
        self.model_outputs = [output_key]
        self.per_atom_output_key = per_atom_output_key
        if per_atom_output_key is not None:
            self.model_outputs.append(per_atom_output_key)
        self.n_out = n_out
        if aggregation_mode is None and per_atom_output_key is None:
            raise ValueError("If `aggregation_mode` is None, `per_atom_output_key` needs to be set,"
                             " since no accumulated output will be returned!")
        self.outnet = build_mlp(n_in=n_in, n_out=n_out, n_hidden=n_hidden, n_layers=n_layers,
                                activation=activation)
        self.aggregation_mode = aggregation_mode
        self.n_molecules_key = n_molecules_key
        self._head_act = 0
        self._fused_head = self._head_fusable()

    def __setstate__(self, state):
        # a model pickled by the REFERENCE (torch.save(model), task.py:300) unpickled onto this class after
        # install(fused_head=True) never ran __init__: fill what the reference's Atomwise does not carry
        super().__setstate__(state)
        if "n_molecules_key" not in self.__dict__:
            self.n_molecules_key = "_n_molecules"
        if "_fused_head" not in self.__dict__ or "_head_act" not in self.__dict__:
            self._head_act = 0
            self._fused_head = self._head_fusable()

    def _head_fusable(self) -> bool:
        """True when the head is the default 2-layer / width-1 MLP the fused HIP kernel covers."""
        if self.aggregation_mode is None or self.n_out != 1:
            return False
        net = self.outnet
        if not (isinstance(net, nn.Sequential) and len(net) == 2 and all(isinstance(l, Dense) for l in net)):
            return False
        act = activation_id(net[0].activation)
        if act is None or act == _lib.SPK_ACT_NONE or activation_id(net[1].activation) != _lib.SPK_ACT_NONE:
            return False
        if net[1].out_features != 1 or net[0].bias is None or net[1].bias is None:
            return False
        if not bool(_lib.lib().spk_atomwise_supported(int(net[0].in_features), int(net[0].out_features), int(act))):
            return False
        self._head_act = int(act)
        return True

    def _n_molecules(self, inputs: Dict[str, torch.Tensor], idx_m: torch.Tensor) -> int:
        # the reference reads int(idx_m[-1]) + 1 (a device sync, atomwise.py:80); a host-side molecule count in the
        # batch dict (python int or CPU tensor under `n_molecules_key`) avoids it when present
        if self.n_molecules_key in inputs:
            return int(inputs[self.n_molecules_key])
        return int(idx_m[-1]) + 1

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = inputs["scalar_representation"]
        if self._fused_head and not self.training and x.dim() == 2 and not use_aten(x):
            idx_m = inputs[properties.idx_m]
            maxm = self._n_molecules(inputs, idx_m)
            l0 = self.outnet[0]
            l1 = self.outnet[1]
            y, y_atom = torch.ops.spk_hip.atomwise(x, l0.weight, l0.bias, l1.weight, l1.bias, idx_m, maxm, self._head_act)
            if self.per_atom_output_key is not None:
                inputs[self.per_atom_output_key] = y_atom
            if self.aggregation_mode == "avg":
                y = y / inputs[properties.n_atoms]
            inputs[self.output_key] = y
            return inputs
        y = self.outnet(x)
        if self.per_atom_output_key is not None:
            inputs[self.per_atom_output_key] = y
        if self.aggregation_mode is not None:
            idx_m = inputs[properties.idx_m]
            maxm = self._n_molecules(inputs, idx_m)
            y = scatter_add(y, idx_m, dim_size=maxm)
            y = torch.squeeze(y, -1)
            if self.aggregation_mode == "avg":
                y = y / inputs[properties.n_atoms]
        inputs[self.output_key] = y
        return inputs


class Forces(nn.Module):
    """forces = -dE/dR (and stress = dE/dstrain / volume) by autograd, ``create_graph = training``, like the reference."""

    def __init__(self, calc_forces: bool = True, calc_stress: bool = False, energy_key: str = properties.energy,
                 force_key: str = properties.forces, stress_key: str = properties.stress):
        super().__init__()
        self.calc_forces = calc_forces
        self.calc_stress = calc_stress
        self.energy_key = energy_key
        self.force_key = force_key
        self.stress_key = stress_key
        self.model_outputs = []
        if calc_forces:
            self.model_outputs.append(force_key)
        if calc_stress:
            self.model_outputs.append(stress_key)
        self.required_derivatives = []
        if calc_forces:
            self.required_derivatives.append(properties.R)
        if calc_stress:
            self.required_derivatives.append(properties.strain)

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        Epred = inputs[self.energy_key]
        go: List[Optional[torch.Tensor]] = [torch.ones_like(Epred)]
        grads = torch.autograd.grad([Epred], [inputs[prop] for prop in self.required_derivatives],
                                    grad_outputs=go, create_graph=self.training)
        if self.calc_forces:
            dEdR = grads[0]
            if dEdR is None:
                dEdR = torch.zeros_like(inputs[properties.R])
            inputs[self.force_key] = -dEdR
        if self.calc_stress:
            stress = grads[-1]
            if stress is None:
                stress = torch.zeros_like(inputs[properties.cell])
            cell = inputs[properties.cell]
            volume = torch.sum(cell[:, 0, :] * torch.cross(cell[:, 1, :], cell[:, 2, :], dim=1), dim=1, keepdim=True)[:, :, None]
            inputs[self.stress_key] = stress / volume
        return inputs
