"""Mirror of ``schnetpack.representation.painn`` (representation/painn.py:14-256) on the gfx950
kernels: same class names, constructor signatures, attributes and ``state_dict`` keys.

Eval mode: ``ops.PaiNNFn`` -- context nets and mixing Dense layers on the fp32 MFMA kernel, the
equivariant message as one fused row kernel per interaction (filters recomputed in registers;
the reference's [E, 1, 3F n_int] filter tensor, painn.py:232, never exists), first-order backward
w.r.t. ``_Rij``.  Training mode: differentiable primitive path.
"""
import ctypes
import os
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from .. import properties
from ..nn import Dense, replicate_module, scatter_add
from ..nn.base import activation_id

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

    def forward(self, q, mu, Wij, dir_ij, idx_i, idx_j, n_atoms: int):
        x = self.interatomic_context_net(q)
        xj = ops.gather(x, idx_j, 0)
        muj = ops.gather(mu, idx_j, 0)
        x = Wij * xj
        dq, dmuR, dmumu = torch.split(x, self.n_atom_basis, dim=-1)
        dq = scatter_add(dq, idx_i, dim_size=n_atoms)
        dmu = dmuR * dir_ij[..., None] + dmumu * muj
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

    def forward(self, q: torch.Tensor, mu: torch.Tensor):
        mu_mix = self.mu_channel_mix(mu)
        mu_V, mu_W = torch.split(mu_mix, self.n_atom_basis, dim=-1)
        mu_Vn = torch.sqrt(torch.sum(mu_V ** 2, dim=-2, keepdim=True) + self.epsilon)
        ctx = torch.cat([q, mu_Vn], dim=-1)
        x = self.intraatomic_context_net(ctx)
        dq_intra, dmu_intra, dqmu_intra = torch.split(x, self.n_atom_basis, dim=-1)
        dmu_intra = dmu_intra * mu_W
        dqmu_intra = dqmu_intra * torch.sum(mu_V * mu_W, dim=1, keepdim=True)
        return q + dq_intra + dqmu_intra, mu + dmu_intra


class PaiNN(nn.Module):
    """PaiNN representation; see the reference docstring (painn.py:120-157) for arguments."""

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
        self._activation = activation
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

    def _act(self):
        # instances restored from reference pickles never ran this __init__
        act = getattr(self, "_activation", None)
        if act is None and len(self.interactions) > 0:
            act = self.interactions[0].interatomic_context_net[0].activation
        return act

    def _eps(self) -> float:
        eps = getattr(self, "epsilon", None)
        if eps is None:
            eps = self.mixing[0].epsilon if len(self.mixing) > 0 else 1e-8
        return float(eps)

    def _fusable(self) -> bool:
        return (activation_id(self._act()) == _lib.SPK_ACT_SILU
                and hasattr(self.radial_basis, "kernel_args")
                and not getattr(self.radial_basis, "trainable", False)
                and hasattr(self.cutoff_fn, "cutoff_value"))

    def _model_struct(self):
        L = self.n_interactions
        Fd = self.n_atom_basis
        params = list(self.filter_net.parameters()) + [p for m in list(self.interactions) + list(self.mixing) for p in m.parameters()]
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.get("_struct_cache")
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        arr = (_lib.PainnLayerT * max(L, 1))()
        keep = []
        fw = self.filter_net.weight.detach().contiguous()
        fb = self.filter_net.bias.detach().contiguous()
        keep += [fw, fb]
        n_rbf = fw.shape[1]
        for l in range(L):
            it, mx = self.interactions[l], self.mixing[l]
            row0 = 0 if self.share_filters else 3 * Fd * l
            ts = {
                "ctx_w1": it.interatomic_context_net[0].weight, "ctx_b1": it.interatomic_context_net[0].bias,
                "ctx_w2": it.interatomic_context_net[1].weight, "ctx_b2": it.interatomic_context_net[1].bias,
                "mix_w": mx.mu_channel_mix.weight,
                "ictx_w1": mx.intraatomic_context_net[0].weight, "ictx_b1": mx.intraatomic_context_net[0].bias,
                "ictx_w2": mx.intraatomic_context_net[1].weight, "ictx_b2": mx.intraatomic_context_net[1].bias,
            }
            for name, t in ts.items():
                t = t.detach().contiguous()
                keep.append(t)
                setattr(arr[l], name, _lib.fptr(t))
                if name in ("ctx_w1", "ctx_w2", "mix_w", "ictx_w1", "ictx_w2"):
                    tt = t.t().contiguous()  # [in, out]: coalesced weight reads in the forward chains
                    keep.append(tt)
                    setattr(arr[l], name + "T", _lib.fptr(tt))
            arr[l].filt_w = ctypes.c_void_p(fw.data_ptr() + 4 * row0 * n_rbf)
            arr[l].filt_b = ctypes.c_void_p(fb.data_ptr() + 4 * row0)
        ms = _lib.PainnT(Fd, L, self._eps(), 0, ctypes.cast(arr, ctypes.POINTER(_lib.PainnLayerT)), None)
        keep.append(arr)
        # packed images of the atom-wise weights for the fused Dense chains (0 floats: shapes without one)
        n_pack = int(_lib.lib().spk_painn_packed_floats(ctypes.byref(ms))) if L > 0 else 0
        if n_pack > 0 and not os.environ.get("SPK_NO_PACK"):
            dev = next(self.parameters()).device
            wpack = torch.empty(n_pack, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().spk_painn_pack_weights_f32(ctypes.byref(ms), _lib.fptr(wpack), _lib.stream()))
            ms.wpack = _lib.fptr(wpack)
            keep.append(wpack)
        self.__dict__["_struct_cache"] = (key, ms, keep)
        return ms, keep

    def forward(self, inputs: Dict[str, torch.Tensor]):
        atomic_numbers = inputs[properties.Z]
        r_ij = inputs[properties.Rij]
        idx_i = inputs[properties.idx_i]
        idx_j = inputs[properties.idx_j]
        n_atoms = atomic_numbers.shape[0]
        ops._check_float(r_ij, "PaiNN")

        q = self.embedding(atomic_numbers)
        for embedding in self.electronic_embeddings:
            q = q + embedding(q, inputs)

        if not self.training and self._fusable():
            plan = ops.edge_plan(idx_i, idx_j, n_atoms, r_ij)
            if plan.filter_pairs is None and plan.n_edges >= (1 << 19):
                # large lists: tells the message dispatch whether the list carries a skin (one sync per list)
                plan.decide_filter(r_ij, self.cutoff_fn.cutoff_value())
            ms, keep = self._model_struct()
            rb_args = self.radial_basis.kernel_args(self.cutoff_fn.cutoff_value())
            # eval path: geometry gradients only (embedding / weights are not differentiated)
            q, mu = ops.PaiNNFn.apply(q.detach(), r_ij, plan, rb_args, ms, keep)
        else:
            d_ij = torch.norm(r_ij, dim=1, keepdim=True)
            dir_ij = r_ij / d_ij
            phi_ij = self.radial_basis(d_ij)
            fcut = self.cutoff_fn(d_ij)
            filters = self.filter_net(phi_ij) * fcut[..., None]
            if self.share_filters:
                filter_list = [filters] * self.n_interactions
            else:
                filter_list = torch.split(filters, 3 * self.n_atom_basis, dim=-1)
            q = q.unsqueeze(1)
            qs = q.shape
            mu = torch.zeros((qs[0], 3, qs[2]), device=q.device)
            for i, (interaction, mixing) in enumerate(zip(self.interactions, self.mixing)):
                q, mu = interaction(q, mu, filter_list[i], dir_ij, idx_i, idx_j, n_atoms)
                q, mu = mixing(q, mu)
            q = q.squeeze(1)

        inputs["scalar_representation"] = q
        inputs["vector_representation"] = mu
        return inputs
