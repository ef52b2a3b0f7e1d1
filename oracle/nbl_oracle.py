"""CPU restatement of the reference's neighbour-list semantics.  TEST INFRASTRUCTURE ONLY -- the
product path (schnetpack_amd.neighborlist -> libspk_hip.so) never imports this module.

Follows ``TorchNeighborList`` (transform/neighborlist.py:438-553, itself after TorchANI's aev.py):
brute force over all atom pairs and all cell shifts that can bring an image within the cutoff.
Pinned against the live reference class (tests/test_oracle_vs_reference.py) and against the
reference's own precomputed Argon vectors (tests/conftest.py:192-447 -> tests/golden/nbl_argon.npz).

The order of pairs inside a row is implementation defined in the reference (``torch.argsort`` of
idx_i, :448) and its own test compares after a canonical sort (tests/data/test_transforms.py:53-104);
``canonical_order`` is that sort with the integer shifts as the tie-breaker.
"""
from typing import Optional, Tuple

import torch

Tensor = torch.Tensor


def half_space_shifts(cell: Tensor, pbc: Tensor, cutoff: float) -> Tensor:
    """Integer shift vectors S != 0 of one half space (S and -S never both) that can hold images
    within ``cutoff`` (transform/neighborlist.py:509-553): n_k = ceil(cutoff * |k-th row of the
    reciprocal cell|) repeats along every periodic axis."""
    recip = torch.linalg.inv(cell).t()
    n = torch.ceil(cutoff * torch.linalg.norm(recip, dim=1)).long()
    n = torch.where(pbc.bool(), n, torch.zeros_like(n))
    rng = [torch.arange(-int(k), int(k) + 1) for k in n]
    S = torch.cartesian_prod(*rng)
    # lexicographically positive half: first non-zero component > 0
    first = torch.where(S[:, 0] != 0, S[:, 0], torch.where(S[:, 1] != 0, S[:, 1], S[:, 2]))
    return S[first > 0]


def neighbor_list(R: Tensor, cell: Optional[Tensor], pbc: Optional[Tensor], cutoff: float
                  ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(idx_i, idx_j, S, offsets) of one system: every directed pair with
    ``|R_j - R_i + S.cell| < cutoff`` (strict, :492-493), ``offsets = S @ cell`` (:457), rows in
    canonical order."""
    n = R.shape[0]
    ar = torch.arange(n)
    periodic = pbc is not None and bool(torch.any(pbc))
    if periodic:
        Sh = half_space_shifts(cell.to(R.dtype), pbc, cutoff)
    else:
        Sh = torch.zeros(0, 3, dtype=torch.long)
        cell = torch.zeros(3, 3, dtype=R.dtype) if cell is None else cell
    cellf = cell.to(R.dtype)
    # central cell: unordered pairs i < j (:473-475); shifted cells: all ordered (i, j) (:480-484)
    pi0, pj0 = torch.combinations(ar).unbind(-1) if n > 1 else (ar[:0], ar[:0])
    s_idx, pi1, pj1 = torch.cartesian_prod(torch.arange(Sh.shape[0]), ar, ar).unbind(-1) if Sh.shape[0] else (ar[:0], ar[:0], ar[:0])
    S_all = torch.cat([torch.zeros(pi0.shape[0], 3, dtype=torch.long), Sh[s_idx]])
    pi = torch.cat([pi0, pi1])
    pj = torch.cat([pj0, pj1])
    vec = R[pi] - R[pj] + S_all.to(R.dtype) @ cellf                 # :488-489
    keep = torch.linalg.norm(vec, dim=1) < cutoff                    # :492-493
    pi, pj, S_all = pi[keep], pj[keep], S_all[keep]
    # both directions (:441-452): (i<-j, -S) and (j<-i, +S)
    idx_i = torch.cat([pi, pj])
    idx_j = torch.cat([pj, pi])
    S = torch.cat([-S_all, S_all])
    order = canonical_order(idx_i, idx_j, S)
    idx_i, idx_j, S = idx_i[order], idx_j[order], S[order]
    return idx_i, idx_j, S, S.to(R.dtype) @ cellf


def canonical_order(idx_i: Tensor, idx_j: Tensor, S: Tensor) -> Tensor:
    """Permutation sorting pairs by (i, j, Sx, Sy, Sz)."""
    if idx_i.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    smin = S.min()
    span = int(S.max() - smin) + 1
    n = int(max(idx_i.max(), idx_j.max())) + 1
    key = (((idx_i * n + idx_j) * span + (S[:, 0] - smin)) * span + (S[:, 1] - smin)) * span + (S[:, 2] - smin)
    return torch.argsort(key)


def batch_neighbor_list(R: Tensor, idx_m: Tensor, cells: Optional[Tensor], pbcs: Optional[Tensor], cutoff: float):
    """Per-system lists concatenated with the atom offset of each system added to the indices
    (what ``_atoms_collate_fn`` does, data/loader.py:35-46).  cells [M,3,3], pbcs [M,3]."""
    n_sys = int(idx_m.max()) + 1 if idx_m.numel() else 0
    out_i, out_j, out_S, out_o = [], [], [], []
    for m in range(n_sys):
        sel = torch.nonzero(idx_m == m).flatten()
        if sel.numel() == 0:
            continue
        a0 = int(sel[0])
        i, j, S, o = neighbor_list(R[sel], None if cells is None else cells[m], None if pbcs is None else pbcs[m], cutoff)
        out_i.append(i + a0); out_j.append(j + a0); out_S.append(S); out_o.append(o)
    if not out_i:
        z = torch.zeros(0, dtype=torch.long)
        return z, z, torch.zeros(0, 3, dtype=torch.long), torch.zeros(0, 3, dtype=R.dtype)
    return torch.cat(out_i), torch.cat(out_j), torch.cat(out_S), torch.cat(out_o)
