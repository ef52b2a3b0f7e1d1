"""Mirror of ``schnetpack.representation.painn`` (representation/painn.py:14-256) on the gfx950
kernels: same class names, constructor signatures, attributes and ``state_dict`` keys.

Eval mode: ONE operator, ``torch.ops.spk_hip.painn`` -- context nets and mixing Dense layers on the fp32 MFMA
kernels, the equivariant message as one fused kernel per interaction (filters recomputed in registers; the
reference's [E, 1, 3F n_int] filter tensor, painn.py:232, never exists), first-order backward w.r.t. ``_Rij`` and
the embedding rows.  Training mode: differentiable primitive path.  Both are TorchScript-able.
"""
from typing import Callable, Dict, Final, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .. import properties
from .. import torchops  # noqa: F401  (registers torch.ops.spk_hip)
from ..nn import Dense, replicate_module, scatter_add
from ..nn.base import activation_id
from ..nn.fallback import note_fallback, use_aten

__all__ = ["PaiNN", "PaiNNInteraction", "PaiNNMixing"]


class PaiNNInteraction(nn.Module):
    r"""PaiNN interaction block; ``forward(q, mu, Wij, dir_ij, idx_i, idx_j, n_atoms)`` keeps the
    reference's materialised-filter signature (painn.py:31-40)."""

    def __init__(self, n_atom_basis: int, activation: Callable):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.interatomic_context_net = nn.Sequential(
            Dense(n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None),
        )

    def fused_weights(self) -> List[torch.Tensor]:
        c0 = self.interatomic_context_net[0]
        c1 = self.interatomic_context_net[1]
        b0, b1 = c0.bias, c1.bias
        assert b0 is not None and b1 is not None
        return [c0.weight, b0, c1.weight, b1]

    def forward(self, q: torch.Tensor, mu: torch.Tensor, Wij: torch.Tensor, dir_ij: torch.Tensor,
                idx_i: torch.Tensor, idx_j: torch.Tensor, n_atoms: int):
        x = self.interatomic_context_net(q)
        if use_aten(x):          # host / non-float32 tensors: the reference's algebra (painn.py:54-66)
            note_fallback()
            xj = x[idx_j]
            muj = mu[idx_j]
            x = Wij * xj
            dq, dmuR, dmumu = torch.split(x, self.n_atom_basis, dim=-1)
            dq = scatter_add(dq, idx_i, dim_size=n_atoms)
            dmu = dmuR * dir_ij[..., None] + dmumu * muj
            dmu = scatter_add(dmu, idx_i, dim_size=n_atoms)
            return q + dq, mu + dmu
        n3 = x.shape[-1]
        # x = Wij * x[idx_j]: one gather-multiply (the filters carry one row per pair: no index on that side)
        x = torch.ops.spk_hip.edge_mul(Wij.reshape(-1, n3), x.reshape(-1, n3), None, idx_j)
        dq, dmuR, dmumu = torch.split(x, self.n_atom_basis, dim=-1)
        dq = scatter_add(dq, idx_i, dim_size=n_atoms).unsqueeze(1)
        muj = torch.ops.spk_hip.gather(mu, idx_j, 0)
        # dmu = dmuR * dir_ij[..., None] + dmumu * muj  (3-vector products that read the halves of the split in place)
        dmu = torch.ops.spk_hip.vec3(2, dmuR, dir_ij) + torch.ops.spk_hip.vec3(0, muj, dmumu)
        dmu = scatter_add(dmu, idx_i, dim_size=n_atoms)
        return q + dq, mu + dmu


class PaiNNMixing(nn.Module):
    r"""PaiNN intra-atomic mixing block (painn.py:70-117)."""

    def __init__(self, n_atom_basis: int, activation: Callable, epsilon: float = 1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.intraatomic_context_net = nn.Sequential(
            Dense(2 * n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None),
        )
        self.mu_channel_mix = Dense(n_atom_basis, 2 * n_atom_basis, activation=None, bias=False)
        self.epsilon = epsilon

    def fused_weights(self) -> List[torch.Tensor]:
        i0 = self.intraatomic_context_net[0]
        i1 = self.intraatomic_context_net[1]
        b0, b1 = i0.bias, i1.bias
        assert b0 is not None and b1 is not None
        return [self.mu_channel_mix.weight, i0.weight, b0, i1.weight, b1]

    def forward(self, q: torch.Tensor, mu: torch.Tensor):
        mu_mix = self.mu_channel_mix(mu)
        mu_V, mu_W = torch.split(mu_mix, self.n_atom_basis, dim=-1)
        if use_aten(mu_mix):     # host / non-float32 tensors: the reference's algebra (painn.py:103-116)
            note_fallback()
            mu_Vn = torch.sqrt(torch.sum(mu_V ** 2, dim=-2, keepdim=True) + self.epsilon)
            ctx = torch.cat([q, mu_Vn], dim=-1)
            x = self.intraatomic_context_net(ctx)
            dq_intra, dmu_intra, dqmu_intra = torch.split(x, self.n_atom_basis, dim=-1)
            dmu_intra = dmu_intra * mu_W
            dqmu_intra = dqmu_intra * torch.sum(mu_V * mu_W, dim=1, keepdim=True)
            return q + dq_intra + dqmu_intra, mu + dmu_intra
        # sum(mu_V ** 2, dim=-2, keepdim=True), dmu_intra * mu_W, sum(mu_V * mu_W, dim=1, keepdim=True): 3-vector products
        # (vec3 codes: 0 = V * s, 1 = sum over the Cartesian axis of A * B) on the halves of mu_mix as they lie
        mu_Vn = torch.sqrt(torch.ops.spk_hip.vec3(1, mu_V, mu_V) + self.epsilon)
        ctx = torch.cat([q, mu_Vn], dim=-1)
        x = self.intraatomic_context_net(ctx)
        dq_intra, dmu_intra, dqmu_intra = torch.split(x, self.n_atom_basis, dim=-1)
        dmu_intra = torch.ops.spk_hip.vec3(0, mu_W, dmu_intra)
        dqmu_intra = dqmu_intra * torch.ops.spk_hip.vec3(1, mu_V, mu_W)
        return q + dq_intra + dqmu_intra, mu + dmu_intra


class PaiNN(nn.Module):
    """PaiNN representation; see the reference docstring (painn.py:120-157) for arguments."""

    _fused: Final[bool]

    def __init__(self, n_atom_basis: int, n_interactions: int, radial_basis: nn.Module,
                 cutoff_fn: Optional[Callable] = None, activation: Optional[Callable] = F.silu,
                 shared_interactions: bool = False, shared_filters: bool = False,
                 epsilon: float = 1e-8, nuclear_embedding: Optional[nn.Module] = None,
                 electronic_embeddings: Optional[List] = None):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.n_interactions = n_interactions
        self.cutoff_fn = cutoff_fn
        self.cutoff = cutoff_fn.cutoff
        self.radial_basis = radial_basis
        self.epsilon = epsilon
        if nuclear_embedding is None:
            nuclear_embedding = nn.Embedding(100, n_atom_basis)
        self.embedding = nuclear_embedding
        if electronic_embeddings is None:
            electronic_embeddings = []
        self.electronic_embeddings = nn.ModuleList(electronic_embeddings)
        self.share_filters = shared_filters
        if shared_filters:
            self.filter_net = Dense(self.radial_basis.n_rbf, 3 * n_atom_basis, activation=None)
        else:
            self.filter_net = Dense(self.radial_basis.n_rbf, self.n_interactions * n_atom_basis * 3,
                                    activation=None)
        self.interactions = replicate_module(
            lambda: PaiNNInteraction(n_atom_basis=self.n_atom_basis, activation=activation),
            self.n_interactions, shared_interactions)
        self.mixing = replicate_module(
            lambda: PaiNNMixing(n_atom_basis=self.n_atom_basis, activation=activation, epsilon=epsilon),
            self.n_interactions, shared_interactions)
        self._fused = self._fusable()

    def __setstate__(self, state):
        # instances restored from reference pickles never ran this __init__
        super().__setstate__(state)
        if not isinstance(self.__dict__.get("_modules", {}).get("electronic_embeddings"), nn.ModuleList):
            self.electronic_embeddings = nn.ModuleList(self.__dict__.pop("electronic_embeddings", None) or [])
        if "epsilon" not in self.__dict__:
            self.epsilon = float(self.mixing[0].epsilon) if len(self.mixing) > 0 else 1e-8
        if "_fused" not in self.__dict__:
            self._fused = self._fusable()

    def _eps(self) -> float:
        return float(self.epsilon)

    def _fusable(self) -> bool:
        """The one-operator eval path covers SiLU context nets, the mirrored radial bases (trainable or not: in eval mode their parameters are plain operands) and cosine
        cutoff within the kernels' shape limits (spk_painn.hip: n_atom_basis <= 1024, n_rbf <= 256)."""
        if len(self.interactions) == 0:
            return False
        acts = [self.interactions[0].interatomic_context_net[0].activation, self.mixing[0].intraatomic_context_net[0].activation]
        n_rbf = int(getattr(self.radial_basis, "n_rbf", 0))
        return (all(activation_id(a) == _lib.SPK_ACT_SILU for a in acts)
                and hasattr(self.radial_basis, "kernel_params")
                and hasattr(self.cutoff_fn, "cutoff_value")
                and self.n_atom_basis <= 1024 and 1 <= n_rbf <= 256)

    def interaction_weights(self) -> List[torch.Tensor]:
        """Nine tensors per interaction in the order of spk_painn_layer_t (include/spk_hip.h), then filter_net.{weight, bias}."""
        ws: List[torch.Tensor] = []
        for interaction, mixing in zip(self.interactions, self.mixing):
            ws += interaction.fused_weights()
            ws += mixing.fused_weights()
        fb = self.filter_net.bias
        assert fb is not None
        ws.append(self.filter_net.weight)
        ws.append(fb)
        return ws

    def embed(self, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        q = self.embedding(inputs[properties.Z])
        for embedding in self.electronic_embeddings:
            q = q + embedding(q, inputs)
        return q

    def forward(self, inputs: Dict[str, torch.Tensor]):
        atomic_numbers = inputs[properties.Z]
        r_ij = inputs[properties.Rij]
        idx_i = inputs[properties.idx_i]
        idx_j = inputs[properties.idx_j]
        n_atoms = atomic_numbers.shape[0]

        q = self.embedding(atomic_numbers)
        for embedding in self.electronic_embeddings:
            q = q + embedding(q, inputs)

        aten = use_aten(r_ij) or use_aten(q)
        if self._fused and not self.training and not aten:
            ws = self.interaction_weights()
            kind, p0, p1 = self.radial_basis.kernel_params()
            q, mu = torch.ops.spk_hip.painn(q, r_ij, idx_i, idx_j, ws, self.share_filters, self.epsilon, kind, p0, p1,
                                            self.cutoff_fn.cutoff_value())
        else:
            d_ij = torch.norm(r_ij, dim=1, keepdim=True) if aten else torch.ops.spk_hip.edge_norm(r_ij).unsqueeze(1)
            dir_ij = r_ij / d_ij
            phi_ij = self.radial_basis(d_ij)
            fcut = self.cutoff_fn(d_ij)
            if aten:
                filters = self.filter_net(phi_ij) * fcut[..., None]
            else:
                filters = torch.ops.spk_hip.rowscale(self.filter_net(phi_ij), fcut)    # filter_net(phi_ij) * fcut[..., None]
            if self.share_filters:
                filter_list = [filters] * self.n_interactions
            else:
                filter_list = torch.split(filters, 3 * self.n_atom_basis, dim=-1)
            q = q.unsqueeze(1)
            qs = q.shape
            mu = torch.zeros((qs[0], 3, qs[2]), device=q.device, dtype=q.dtype if aten else torch.float32)
            for i, (interaction, mixing) in enumerate(zip(self.interactions, self.mixing)):
                q, mu = interaction(q, mu, filter_list[i], dir_ij, idx_i, idx_j, n_atoms)
                q, mu = mixing(q, mu)
            q = q.squeeze(1)

        inputs["scalar_representation"] = q
        inputs["vector_representation"] = mu
        return inputs
