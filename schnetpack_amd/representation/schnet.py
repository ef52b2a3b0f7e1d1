"""Mirror of ``schnetpack.representation.schnet`` (representation/schnet.py:14-173) on the
gfx950 kernels: same class names, constructor signatures, attributes and ``state_dict`` keys.

``SchNet.forward`` in eval mode runs the whole representation as one fused HIP call
(``ops.SchNetFn``: in2f -> fused cfconv (RBF x cutoff x filter MLP x gather x segmented sum, no
[E, F] tensor) -> f2out, for every interaction) and its first-order backward w.r.t. ``_Rij`` --
what ``Forces`` asks for.  In training mode (force loss => double backward) it runs the
differentiable primitive path: HIP Dense / gather / scatter_add with torch elementwise algebra.
"""
import ctypes
import os
from typing import Callable, Dict, List, Optional, Union

import torch
from torch import nn

from .. import _lib, ops
from .. import properties
from ..nn import Dense, scatter_add
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

    def forward(self, x: torch.Tensor, f_ij: torch.Tensor, idx_i: torch.Tensor,
                idx_j: torch.Tensor, rcut_ij: torch.Tensor):
        x = self.in2f(x)
        Wij = self.filter_network(f_ij)
        Wij = Wij * rcut_ij[:, None]
        x_j = ops.gather(x, idx_j, 0)
        x_ij = x_j * Wij
        x = scatter_add(x_ij, idx_i, dim_size=x.shape[0])
        return self.f2out(x)


class SchNet(nn.Module):
    """SchNet representation; see the reference docstring (schnet.py:73-116) for arguments."""

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
        self._activation = activation
        self.interactions = replicate_module(
            lambda: SchNetInteraction(n_atom_basis=self.n_atom_basis, n_rbf=self.radial_basis.n_rbf,
                                      n_filters=self.n_filters, activation=activation),
            n_interactions, shared_interactions)

    # -- fused eval path ---------------------------------------------------------------
    def _act(self):
        # instances restored from reference pickles never ran this __init__
        act = getattr(self, "_activation", None)
        if act is None and len(self.interactions) > 0:
            act = self.interactions[0].filter_network[0].activation
        return act

    def _fusable(self) -> bool:
        return (activation_id(self._act()) == _lib.SPK_ACT_SSP
                and hasattr(self.radial_basis, "kernel_args")
                and not getattr(self.radial_basis, "trainable", False)
                and hasattr(self.cutoff_fn, "cutoff_value"))

    def _model_struct(self):
        """ctypes parameter block (device pointers of the state_dict tensors + cached transposed
        copies of the atom-wise weights for coalesced reads; rebuilt when a parameter changes)."""
        L = len(self.interactions)
        params = [p for it in self.interactions for p in it.parameters()]
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.get("_struct_cache")
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        arr = (_lib.SchnetLayerT * max(L, 1))()
        keep = []
        for l, it in enumerate(self.interactions):
            ts = [it.in2f.weight, it.filter_network[0].weight, it.filter_network[0].bias,
                  it.filter_network[1].weight, it.filter_network[1].bias, it.f2out[0].weight,
                  it.f2out[0].bias, it.f2out[1].weight, it.f2out[1].bias]
            ts = [t.detach().contiguous() for t in ts]
            ts += [ts[0].t().contiguous(), ts[5].t().contiguous(), ts[7].t().contiguous()]
            keep.extend(ts)
            for name, t in zip([f[0] for f in _lib.SchnetLayerT._fields_], ts):
                setattr(arr[l], name, _lib.fptr(t))
        ms = _lib.SchnetT(self.n_atom_basis, self.n_filters, L, 0,
                          ctypes.cast(arr, ctypes.POINTER(_lib.SchnetLayerT)), None)
        keep.append(arr)
        # packed images of the atom-wise weights for the fused Dense chains (0 floats: shapes without one)
        n_pack = int(_lib.lib().spk_schnet_packed_floats(ctypes.byref(ms))) if L > 0 else 0
        if n_pack > 0 and not os.environ.get("SPK_NO_PACK"):
            wpack = torch.empty(n_pack, dtype=torch.float32, device=params[0].device)
            with torch.cuda.device(wpack.device):
                _lib.check(_lib.lib().spk_schnet_pack_weights_f32(ctypes.byref(ms), _lib.fptr(wpack), _lib.stream()))
            ms.wpack = _lib.fptr(wpack)
            keep.append(wpack)
        self.__dict__["_struct_cache"] = (key, ms, keep)
        return ms, keep

    def forward(self, inputs: Dict[str, torch.Tensor]):
        atomic_numbers = inputs[properties.Z]
        r_ij = inputs[properties.Rij]
        idx_i = inputs[properties.idx_i]
        idx_j = inputs[properties.idx_j]
        ops._check_float(r_ij, "SchNet")

        x = self.embedding(atomic_numbers)
        for embedding in self.electronic_embeddings:
            x = x + embedding(x, inputs)

        if not self.training and self._fusable():
            plan = ops.edge_plan(idx_i, idx_j, x.shape[0], r_ij)
            if plan.filter_pairs is None:
                plan.decide_filter(r_ij, self.cutoff_fn.cutoff_value())
            ms, keep = self._model_struct()
            rb_args = self.radial_basis.kernel_args(self.cutoff_fn.cutoff_value())
            # eval path: geometry gradients only (embedding / weights are not differentiated)
            x = ops.SchNetFn.apply(x.detach(), r_ij, plan, rb_args, ms, keep)
        else:
            d_ij = torch.norm(r_ij, dim=1)
            f_ij = self.radial_basis(d_ij)
            rcut_ij = self.cutoff_fn(d_ij)
            for interaction in self.interactions:
                v = interaction(x, f_ij, idx_i, idx_j, rcut_ij)
                x = x + v

        inputs["scalar_representation"] = x
        return inputs
