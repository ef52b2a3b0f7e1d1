"""Mirror of ``schnetpack.representation.schnet`` (representation/schnet.py:14-173) on the
gfx950 kernels: same class names, constructor signatures, attributes and ``state_dict`` keys.

``SchNet.forward`` in eval mode runs the whole representation as ONE operator, ``torch.ops.spk_hip.schnet``
(in2f -> fused cfconv (RBF x cutoff x filter MLP x gather x segmented sum, no [E, F] tensor) -> f2out, for every
interaction; one C call forward, one C call for the first-order backward w.r.t. ``_Rij`` and the embedding rows --
what ``Forces`` asks for).  In training mode (force loss => double backward) it runs the differentiable primitive
path: HIP operators that are closed under differentiation (Dense, radial functions, ``rowscale``, ``cfconv``; csrc/spk_train.hip),
so the recorded backward and its backward are HIP launches too.  Both are TorchScript-able
(reference tests/nn/test_schnet.py:83-96).
"""
from typing import Callable, Dict, Final, List, Optional, Union

import torch
from torch import nn

from .. import _lib
from .. import properties
from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from ..nn import Dense, scatter_add
from ..nn.fallback import note_fallback, use_aten
from ..nn import replicate_module
from ..nn.activations import shifted_softplus
from ..nn.base import activation_id

__all__ = ["SchNet", "SchNetInteraction"]


class SchNetInteraction(nn.Module):
    r"""SchNet interaction block (cfconv).  ``forward(x, f_ij, idx_i, idx_j, rcut_ij)`` keeps the
    reference's materialised-input signature (schnet.py:41-48) and runs on HIP primitives."""

    def __init__(self, n_atom_basis: int, n_rbf: int, n_filters: int,
                 activation: Callable = shifted_softplus):
        super().__init__()
        self.in2f = Dense(n_atom_basis, n_filters, bias=False, activation=None)
        self.f2out = nn.Sequential(
            Dense(n_filters, n_atom_basis, activation=activation),
            Dense(n_atom_basis, n_atom_basis, activation=None),
        )
        self.filter_network = nn.Sequential(
            Dense(n_rbf, n_filters, activation=activation), Dense(n_filters, n_filters)
        )

    def fused_weights(self) -> List[torch.Tensor]:
        """The nine tensors of this block in the order of ``spk_schnet_layer_t`` (include/spk_hip.h)."""
        fn0 = self.filter_network[0]
        fn1 = self.filter_network[1]
        o0 = self.f2out[0]
        o1 = self.f2out[1]
        b0, b1, b2, b3 = fn0.bias, fn1.bias, o0.bias, o1.bias
        assert b0 is not None and b1 is not None and b2 is not None and b3 is not None
        return [self.in2f.weight, fn0.weight, b0, fn1.weight, b1, o0.weight, b2, o1.weight, b3]

    def forward(self, x: torch.Tensor, f_ij: torch.Tensor, idx_i: torch.Tensor,
                idx_j: torch.Tensor, rcut_ij: torch.Tensor):
        if use_aten(x):          # host / non-float32 tensors: the reference's algebra (schnet.py:60-67)
            note_fallback()
            x = self.in2f(x)
            Wij = self.filter_network(f_ij)
            Wij = Wij * rcut_ij[:, None]
            x_ij = x[idx_j] * Wij
            x = scatter_add(x_ij, idx_i, dim_size=x.shape[0])
            return self.f2out(x)
        x = self.in2f(x)
        Wij = self.filter_network(f_ij)
        Wij = torch.ops.spk_hip.rowscale(Wij, rcut_ij)                             # Wij * rcut_ij[:, None]
        x = torch.ops.spk_hip.cfconv(x, Wij, idx_i, idx_j, x.shape[0])              # scatter_add(x[idx_j] * Wij, idx_i)
        return self.f2out(x)


class SchNet(nn.Module):
    """SchNet representation; see the reference docstring (schnet.py:73-116) for arguments."""

    _fused: Final[bool]

    def __init__(self, n_atom_basis: int, n_interactions: int, radial_basis: nn.Module,
                 cutoff_fn: Callable, n_filters: int = None, shared_interactions: bool = False,
                 activation: Union[Callable, nn.Module] = shifted_softplus,
                 nuclear_embedding: Optional[nn.Module] = None,
                 electronic_embeddings: Optional[List] = None):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.n_filters = n_filters or self.n_atom_basis
        self.radial_basis = radial_basis
        self.cutoff_fn = cutoff_fn
        self.cutoff = cutoff_fn.cutoff
        if nuclear_embedding is None:
            nuclear_embedding = nn.Embedding(100, n_atom_basis)
        self.embedding = nuclear_embedding
        if electronic_embeddings is None:
            electronic_embeddings = []
        self.electronic_embeddings = nn.ModuleList(electronic_embeddings)
        self.interactions = replicate_module(
            lambda: SchNetInteraction(n_atom_basis=self.n_atom_basis, n_rbf=self.radial_basis.n_rbf,
                                      n_filters=self.n_filters, activation=activation),
            n_interactions, shared_interactions)
        self._fused = self._fusable()

    def __setstate__(self, state):
        # instances restored from reference pickles never ran this __init__
        super().__setstate__(state)
        if not isinstance(self.__dict__.get("_modules", {}).get("electronic_embeddings"), nn.ModuleList):
            self.electronic_embeddings = nn.ModuleList(self.__dict__.pop("electronic_embeddings", None) or [])
        if "_fused" not in self.__dict__:
            self._fused = self._fusable()

    def _fusable(self) -> bool:
        """The one-operator eval path covers ssp filters / output nets, the mirrored radial bases (trainable or not: in eval mode their parameters are plain operands) and
        cosine cutoff, within the kernels' shape limits (spk_cfconv.hip: n_filters % 4, n_rbf <= 256)."""
        if len(self.interactions) == 0:
            return True
        it = self.interactions[0]
        acts = [it.filter_network[0].activation, it.f2out[0].activation]
        n_rbf = int(getattr(self.radial_basis, "n_rbf", 0))
        return (all(activation_id(a) == _lib.SPK_ACT_SSP for a in acts)
                and hasattr(self.radial_basis, "kernel_params")
                and hasattr(self.cutoff_fn, "cutoff_value")
                and self.n_filters % 4 == 0 and self.n_filters <= 1024 and 1 <= n_rbf <= 256)

    def embed(self, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """x_0: nuclear embedding rows plus the electronic embeddings (schnet.py:160-165)."""
        x = self.embedding(inputs[properties.Z])
        for embedding in self.electronic_embeddings:
            x = x + embedding(x, inputs)
        return x

    def interaction_weights(self) -> List[torch.Tensor]:
        """The tensors of all interaction blocks in the order of ``spk_schnet_layer_t`` (what the fused operators take)."""
        ws: List[torch.Tensor] = []
        for interaction in self.interactions:
            ws += interaction.fused_weights()
        return ws

    def forward(self, inputs: Dict[str, torch.Tensor]):
        r_ij = inputs[properties.Rij]
        idx_i = inputs[properties.idx_i]
        idx_j = inputs[properties.idx_j]

        x = self.embed(inputs)

        aten = use_aten(r_ij) or use_aten(x)
        if self._fused and not self.training and not aten:
            ws = self.interaction_weights()
            kind, p0, p1 = self.radial_basis.kernel_params()
            x = torch.ops.spk_hip.schnet(x, r_ij, idx_i, idx_j, ws, self.n_filters, kind, p0, p1, self.cutoff_fn.cutoff_value())
        else:
            d_ij = torch.norm(r_ij, dim=1) if aten else torch.ops.spk_hip.edge_norm(r_ij)
            f_ij = self.radial_basis(d_ij)
            rcut_ij = self.cutoff_fn(d_ij)
            for interaction in self.interactions:
                v = interaction(x, f_ij, idx_i, idx_j, rcut_ij)
                x = x + v

        inputs["scalar_representation"] = x
        return inputs
