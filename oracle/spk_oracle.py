"""CPU oracle for the SchNet / PaiNN message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (plain functional torch on the CPU, fp32 or fp64) of the
algorithm that the reference implements with ``nn.Module`` classes.  It is the checker the
HIP path is compared against; it is never the thing that is shipped or measured (except as
the ``cpu_baseline`` leg of ``bench.py``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning: ``tests/test_oracle_vs_reference.py`` compares every function below against the live
reference modules (imported through ``oracle/refshim.py``) when ``/root/reference`` exists, and
``tests/test_oracle_golden.py`` compares it against the committed fixtures in ``tests/golden``
(generated from the reference by ``oracle/make_golden.py``) plus the reference's own
known-answer tests for RBF / cutoff / shifted-softplus (``tests/nn/test_radial.py``,
``test_cutoff.py``, ``test_activations.py``).

Parameters are passed as flat dicts that use the reference's ``state_dict`` key names
(SURVEY.md §5 "checkpoint"), e.g. ``interactions.0.filter_network.1.weight``.

All path:line citations are into ``/root/reference/src/schnetpack``.
"""
import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
LN2 = math.log(2.0)


# ----------------------------------------------------------------------------- L0 blocks
def scatter_add(x: Tensor, idx: Tensor, dim_size: int, dim: int = 0) -> Tensor:
    """y[..., k, ...] = sum of x[..., e, ...] over e with idx[e] == k  (nn/scatter.py:7-34)."""
    shape = list(x.shape)
    shape[dim] = dim_size
    y = torch.zeros(shape, dtype=x.dtype, device=x.device)
    return y.index_add(dim, idx, x)


def gaussian_rbf(d: Tensor, offsets: Tensor, widths: Tensor) -> Tensor:
    """exp(-0.5/w_k^2 (d-mu_k)^2) on a trailing axis  (nn/radial.py:11-15)."""
    c = -0.5 / (widths * widths)
    delta = d.unsqueeze(-1) - offsets
    return torch.exp(c * delta * delta)


def gaussian_rbf_params(n_rbf: int, cutoff: float, start: float = 0.0) -> Tuple[Tensor, Tensor]:
    """offsets = linspace(start, cutoff, n_rbf); widths = |mu_1 - mu_0|  (nn/radial.py:36-39)."""
    offsets = torch.linspace(start, cutoff, n_rbf)
    widths = torch.abs(offsets[1] - offsets[0]) * torch.ones_like(offsets)
    return offsets, widths


def bessel_rbf(d: Tensor, freqs: Tensor) -> Tensor:
    """sin(k pi d / rc) / d with the d==0 guard  (nn/radial.py:105-110)."""
    s = torch.sin(d.unsqueeze(-1) * freqs)
    safe = torch.where(d == 0, torch.ones_like(d), d)
    return s / safe.unsqueeze(-1)


def bessel_rbf_params(n_rbf: int, cutoff: float) -> Tensor:
    """freqs = k pi / rc, k = 1..n_rbf  (nn/radial.py:102)."""
    return torch.arange(1, n_rbf + 1) * math.pi / cutoff


def cosine_cutoff(d: Tensor, cutoff) -> Tensor:
    """0.5 (cos(pi d / rc) + 1) [d < rc]  (nn/cutoff.py:14-33)."""
    f = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)
    return f * (d < cutoff).to(f.dtype)


def shifted_softplus(x: Tensor) -> Tensor:
    """softplus(x) - ln 2, softplus with beta=1 / threshold=20  (nn/activations.py:9-22)."""
    return torch.nn.functional.softplus(x) - LN2


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def dense(x: Tensor, w: Tensor, b: Optional[Tensor] = None, act=None) -> Tensor:
    """act(x W^T + b)  (nn/base.py:52-55)."""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return act(y) if act is not None else y


def pairwise_vectors(R: Tensor, idx_i: Tensor, idx_j: Tensor, offsets: Tensor) -> Tensor:
    """r_ij = R[j] - R[i] + offsets  (atomistic/distances.py:14-26)."""
    return R[idx_j] - R[idx_i] + offsets


# ----------------------------------------------------------------------------- SchNet
def schnet_interaction(x, f_ij, idx_i, idx_j, rcut_ij, p: Dict[str, Tensor], pre: str) -> Tensor:
    """One cfconv block, returns v (the residual is added by the caller)
    (representation/schnet.py:41-70)."""
    h = dense(x, p[pre + "in2f.weight"])
    W = dense(f_ij, p[pre + "filter_network.0.weight"], p[pre + "filter_network.0.bias"],
              shifted_softplus)
    W = dense(W, p[pre + "filter_network.1.weight"], p[pre + "filter_network.1.bias"])
    W = W * rcut_ij[:, None]
    y = scatter_add(h[idx_j] * W, idx_i, x.shape[0])
    v = dense(y, p[pre + "f2out.0.weight"], p[pre + "f2out.0.bias"], shifted_softplus)
    return dense(v, p[pre + "f2out.1.weight"], p[pre + "f2out.1.bias"])


def _radial(d, p: Dict[str, Tensor]):
    if "radial_basis.freqs" in p:
        return bessel_rbf(d, p["radial_basis.freqs"])
    return gaussian_rbf(d, p["radial_basis.offsets"], p["radial_basis.widths"])


def schnet_representation(Z, r_ij, idx_i, idx_j, p: Dict[str, Tensor], n_interactions: int) -> Tensor:
    """scalar_representation [N, F]  (representation/schnet.py:147-173)."""
    d = torch.sqrt((r_ij * r_ij).sum(dim=1))
    f_ij = _radial(d, p)
    rcut = cosine_cutoff(d, p["cutoff_fn.cutoff"])
    x = p["embedding.weight"][Z]
    for l in range(n_interactions):
        x = x + schnet_interaction(x, f_ij, idx_i, idx_j, rcut, p, "interactions.%d." % l)
    return x


# ----------------------------------------------------------------------------- PaiNN
def painn_interaction(q, mu, Wij, dir_ij, idx_i, idx_j, n_atoms, p, pre: str):
    """q [N,1,F], mu [N,3,F], Wij [E,1,3F] (already x cutoff), dir_ij [E,3]
    (representation/painn.py:31-67)."""
    F = q.shape[-1]
    c = dense(q, p[pre + "interatomic_context_net.0.weight"],
              p[pre + "interatomic_context_net.0.bias"], silu)
    c = dense(c, p[pre + "interatomic_context_net.1.weight"],
              p[pre + "interatomic_context_net.1.bias"])
    m = Wij * c[idx_j]
    m_q, m_R, m_mu = m[..., :F], m[..., F:2 * F], m[..., 2 * F:]
    dq = scatter_add(m_q, idx_i, n_atoms)
    dmu = m_R * dir_ij[..., None] + m_mu * mu[idx_j]
    dmu = scatter_add(dmu, idx_i, n_atoms)
    return q + dq, mu + dmu


def painn_mixing(q, mu, p, pre: str, eps: float = 1e-8):
    """(representation/painn.py:92-117)."""
    F = q.shape[-1]
    mix = dense(mu, p[pre + "mu_channel_mix.weight"])
    V, W = mix[..., :F], mix[..., F:]
    Vn = torch.sqrt((V * V).sum(dim=-2, keepdim=True) + eps)
    ctx = torch.cat([q, Vn], dim=-1)
    a = dense(ctx, p[pre + "intraatomic_context_net.0.weight"],
              p[pre + "intraatomic_context_net.0.bias"], silu)
    a = dense(a, p[pre + "intraatomic_context_net.1.weight"],
              p[pre + "intraatomic_context_net.1.bias"])
    a_q, a_mu, a_qmu = a[..., :F], a[..., F:2 * F], a[..., 2 * F:]
    q = q + a_q + a_qmu * (V * W).sum(dim=1, keepdim=True)
    mu = mu + a_mu * W
    return q, mu


def painn_representation(Z, r_ij, idx_i, idx_j, p, n_interactions: int,
                         shared_filters: bool = False, eps: float = 1e-8):
    """(scalar_representation [N,F], vector_representation [N,3,F])
    (representation/painn.py:207-256)."""
    n_atoms = Z.shape[0]
    d = torch.sqrt((r_ij * r_ij).sum(dim=1, keepdim=True))  # [E,1]
    u = r_ij / d
    phi = _radial(d, p)  # [E,1,n_rbf]
    fcut = cosine_cutoff(d, p["cutoff_fn.cutoff"])
    filters = dense(phi, p["filter_net.weight"], p["filter_net.bias"]) * fcut[..., None]
    q = p["embedding.weight"][Z].unsqueeze(1)
    F = q.shape[-1]
    mu = torch.zeros((n_atoms, 3, F), dtype=q.dtype)
    for l in range(n_interactions):
        Wl = filters if shared_filters else filters[..., 3 * F * l:3 * F * (l + 1)]
        q, mu = painn_interaction(q, mu, Wl, u, idx_i, idx_j, n_atoms, p, "interactions.%d." % l)
        q, mu = painn_mixing(q, mu, p, "mixing.%d." % l, eps)
    return q.squeeze(1), mu


# ----------------------------------------------------------------------------- heads
def atomwise_energy(x: Tensor, idx_m: Tensor, n_mol: int, p: Dict[str, Tensor], pre: str = "",
                    n_layers: int = 2) -> Tensor:
    """Pyramidal MLP (SiLU hidden layers) + sum per molecule
    (atomistic/atomwise.py:69-88, nn/blocks.py:38-76)."""
    y = x
    for k in range(n_layers):
        act = silu if k < n_layers - 1 else None
        y = dense(y, p[pre + "outnet.%d.weight" % k], p[pre + "outnet.%d.bias" % k], act)
    return scatter_add(y, idx_m, n_mol).squeeze(-1)


def energy_and_forces(kind: str, rep_p: Dict[str, Tensor], head_p: Dict[str, Tensor], batch,
                      n_interactions: int, dtype=torch.float32, shared_filters: bool = False,
                      need_rep: bool = False):
    """Full force call: PairwiseDistances -> representation -> Atomwise -> -dE/dR
    (model/base.py:174-190, atomistic/response.py:59-76).  ``batch`` holds Z, R, idx_i, idx_j,
    offsets, idx_m, n_mol.  Returns dict(energy [M], forces [N,3], [scalar, vector])."""
    rep_p = {k: v.to(dtype) if v.is_floating_point() else v for k, v in rep_p.items()}
    head_p = {k: v.to(dtype) for k, v in head_p.items()}
    R = batch["R"].detach().to(dtype).clone().requires_grad_(True)
    offsets = batch["offsets"].to(dtype)
    r_ij = pairwise_vectors(R, batch["idx_i"], batch["idx_j"], offsets)
    out = {}
    if kind == "schnet":
        x = schnet_representation(batch["Z"], r_ij, batch["idx_i"], batch["idx_j"], rep_p,
                                  n_interactions)
    elif kind == "painn":
        x, mu = painn_representation(batch["Z"], r_ij, batch["idx_i"], batch["idx_j"], rep_p,
                                     n_interactions, shared_filters)
        if need_rep:
            out["vector_representation"] = mu.detach()
    else:
        raise ValueError(kind)
    E = atomwise_energy(x, batch["idx_m"], int(batch["n_mol"]), head_p)
    (dEdR,) = torch.autograd.grad([E.sum()], [R])
    out["energy"] = E.detach()
    out["forces"] = -dEdR
    if need_rep:
        out["scalar_representation"] = x.detach()
    return out


# ----------------------------------------------------------------------------- seeded init
def _xavier_uniform(out_f: int, in_f: int, gen=None) -> Tensor:
    """torch.nn.init.xavier_uniform_ on a [out,in] matrix (nn/base.py:27-28,47-50): consumes
    the global RNG exactly like the reference's Dense.reset_parameters."""
    w = torch.empty(out_f, in_f)
    torch.nn.init.xavier_uniform_(w)
    return w


def init_schnet_params(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, n_filters=None,
                       seed=0, radial="gaussian") -> Dict[str, Tensor]:
    """Seeded parameters in the reference's construction order (schnet.py:117-145:
    embedding N(0,1) first, then per interaction in2f, f2out.0, f2out.1, filter_network.0,
    filter_network.1; Dense = xavier_uniform weights, zero bias)."""
    nf = n_filters or n_atom_basis
    torch.manual_seed(seed)
    p = {}
    if radial == "gaussian":
        p["radial_basis.offsets"], p["radial_basis.widths"] = gaussian_rbf_params(n_rbf, cutoff)
    else:
        p["radial_basis.freqs"] = bessel_rbf_params(n_rbf, cutoff).to(torch.float32)
    p["cutoff_fn.cutoff"] = torch.tensor([cutoff], dtype=torch.float32)
    p["embedding.weight"] = torch.randn(100, n_atom_basis)
    for l in range(n_interactions):
        pre = "interactions.%d." % l
        p[pre + "in2f.weight"] = _xavier_uniform(nf, n_atom_basis)
        p[pre + "f2out.0.weight"] = _xavier_uniform(n_atom_basis, nf)
        p[pre + "f2out.0.bias"] = torch.zeros(n_atom_basis)
        p[pre + "f2out.1.weight"] = _xavier_uniform(n_atom_basis, n_atom_basis)
        p[pre + "f2out.1.bias"] = torch.zeros(n_atom_basis)
        p[pre + "filter_network.0.weight"] = _xavier_uniform(nf, n_rbf)
        p[pre + "filter_network.0.bias"] = torch.zeros(nf)
        p[pre + "filter_network.1.weight"] = _xavier_uniform(nf, nf)
        p[pre + "filter_network.1.bias"] = torch.zeros(nf)
    return p


def init_painn_params(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, seed=0,
                      shared_filters=False, radial="gaussian") -> Dict[str, Tensor]:
    """Seeded parameters in the reference's construction order (painn.py:158-205: embedding,
    filter_net, all interactions (context net 0, 1), then all mixing blocks
    (intraatomic_context_net.0, .1, mu_channel_mix))."""
    F = n_atom_basis
    torch.manual_seed(seed)
    p = {}
    if radial == "gaussian":
        p["radial_basis.offsets"], p["radial_basis.widths"] = gaussian_rbf_params(n_rbf, cutoff)
    else:
        p["radial_basis.freqs"] = bessel_rbf_params(n_rbf, cutoff).to(torch.float32)
    p["cutoff_fn.cutoff"] = torch.tensor([cutoff], dtype=torch.float32)
    p["embedding.weight"] = torch.randn(100, F)
    nfo = 3 * F if shared_filters else 3 * F * n_interactions
    p["filter_net.weight"] = _xavier_uniform(nfo, n_rbf)
    p["filter_net.bias"] = torch.zeros(nfo)
    for l in range(n_interactions):
        pre = "interactions.%d.interatomic_context_net." % l
        p[pre + "0.weight"] = _xavier_uniform(F, F)
        p[pre + "0.bias"] = torch.zeros(F)
        p[pre + "1.weight"] = _xavier_uniform(3 * F, F)
        p[pre + "1.bias"] = torch.zeros(3 * F)
    for l in range(n_interactions):
        pre = "mixing.%d." % l
        p[pre + "intraatomic_context_net.0.weight"] = _xavier_uniform(F, 2 * F)
        p[pre + "intraatomic_context_net.0.bias"] = torch.zeros(F)
        p[pre + "intraatomic_context_net.1.weight"] = _xavier_uniform(3 * F, F)
        p[pre + "intraatomic_context_net.1.bias"] = torch.zeros(3 * F)
        p[pre + "mu_channel_mix.weight"] = _xavier_uniform(2 * F, F)
    return p


def init_atomwise_params(n_in=128, n_out=1, n_layers=2, seed=1) -> Dict[str, Tensor]:
    """Pyramidal build_mlp sizes (nn/blocks.py:38-46): n_in -> n_in/2 -> ... -> n_out."""
    torch.manual_seed(seed)
    sizes = []
    c = n_in
    for _ in range(n_layers):
        sizes.append(c)
        c = max(n_out, c // 2)
    sizes.append(n_out)
    p = {}
    for k in range(n_layers):
        p["outnet.%d.weight" % k] = _xavier_uniform(sizes[k + 1], sizes[k])
        p["outnet.%d.bias" % k] = torch.zeros(sizes[k + 1])
    return p
