"""Seeded synthetic inputs for the hot path; no dataset download.

The batch layout is the one the reference's collate function produces
(``data/loader.py:13-58``): atoms of all systems concatenated, ``idx_i/idx_j`` offset by the
cumulative atom count, ``idx_m`` = system index of every atom, ``idx_i`` sorted ascending,
int64 indices.  Neighbour pairs are *full, symmetric* lists (both i<-j and j<-i) like every
neighbour-list back-end of the reference returns (``transform/neighborlist.py:446-456``).
"""
import math
from typing import Dict

import numpy as np
import torch

# 21-atom aspirin geometry (Angstrom) and atomic numbers of
# interfaces/lammps/examples/aspirin/aspirin.data (types 1,2,3 -> Z 6,1,8 per aspirin_md.in:9)
ASPIRIN_Z = [6, 6, 6, 6, 6, 6, 6, 8, 8, 8, 6, 6, 8, 1, 1, 1, 1, 1, 1, 1, 1]
ASPIRIN_R = [
    [7.13448882, 4.01563895, 4.80478211], [5.76264381, 5.95941395, 3.32007110],
    [7.66034484, 4.59207395, 3.69269609], [6.91031682, 5.39396596, 2.85298014],
    [1.96980977, 6.49540496, 5.71966213], [5.84942484, 4.44912893, 5.28437510],
    [5.23844682, 5.47350594, 4.59557810], [5.89789581, 2.72356796, 6.73006105],
    [2.61654782, 5.41777894, 3.53714311], [4.52379882, 4.47091293, 7.33925915],
    [5.39299181, 3.80976295, 6.53798211], [2.87701488, 5.95175993, 4.60229510],
    [4.19533384, 6.28624594, 5.11050910], [4.50619683, 3.81320798, 8.09597421],
    [7.55473471, 3.19750297, 5.39213109], [5.33068982, 6.85571098, 2.65473604],
    [8.80379391, 4.50628096, 3.54379714], [7.23114085, 5.55718595, 1.87585020],
    [2.29106975, 7.48465800, 5.92692810], [0.86951685, 6.48216701, 5.43126610],
    [2.12585187, 6.00320899, 6.69948506],
]
# 9-atom ethanol of tests/testdata/md_ethanol.xyz
ETHANOL_Z = [6, 6, 1, 1, 1, 1, 1, 8, 1]
ETHANOL_R = [
    [-4.92196480914482, 1.53680877549233, -0.06612792847094],
    [-3.41079303549336, 1.45138155063184, -0.14009009720834],
    [-5.22648850340463, 2.28202241947302, 0.66236410391492],
    [-5.34004680800574, 0.57895313793668, 0.22257334141131],
    [-5.33193076526251, 1.80898014947387, -1.03229511269262],
    [-3.00368348713509, 1.18933429199764, 0.83479697695625],
    [-2.99557504133053, 2.41817570143478, -0.41886385105291],
    [-3.07553304550781, 0.47652256654287, -1.09348059854212],
    [-2.13350450471551, 0.40432140701697, -1.15817683431555],
]


def neighbor_pairs_open(R: np.ndarray, cutoff: float):
    """All directed pairs (i, j), i != j, with |R_j - R_i| < cutoff for one non-periodic
    system, sorted by (i, j).  float32 distance test like the reference's torch list
    (``transform/neighborlist.py:495-500``)."""
    R32 = R.astype(np.float32)
    diff = R32[None, :, :] - R32[:, None, :]
    d = np.sqrt((diff * diff).sum(-1, dtype=np.float32))
    mask = d < np.float32(cutoff)
    np.fill_diagonal(mask, False)
    ii, jj = np.nonzero(mask)  # row-major => sorted by (i, j)
    return ii.astype(np.int64), jj.astype(np.int64)


def collate(systems) -> Dict[str, torch.Tensor]:
    """Concatenate systems the way ``_atoms_collate_fn`` does (data/loader.py:13-58)."""
    Z, R, ii, jj, off, idx_m = [], [], [], [], [], []
    start = 0
    for m, s in enumerate(systems):
        n = len(s["Z"])
        Z.append(np.asarray(s["Z"], dtype=np.int64))
        R.append(np.asarray(s["R"], dtype=np.float32))
        ii.append(s["idx_i"] + start)
        jj.append(s["idx_j"] + start)
        off.append(s.get("offsets", np.zeros((len(s["idx_i"]), 3), dtype=np.float32)))
        idx_m.append(np.full(n, m, dtype=np.int64))
        start += n
    batch = {
        "Z": torch.from_numpy(np.concatenate(Z)),
        "R": torch.from_numpy(np.concatenate(R)),
        "idx_i": torch.from_numpy(np.concatenate(ii)),
        "idx_j": torch.from_numpy(np.concatenate(jj)),
        "offsets": torch.from_numpy(np.concatenate(off).astype(np.float32)),
        "idx_m": torch.from_numpy(np.concatenate(idx_m)),
        "n_mol": len(systems),
    }
    return batch


def molecule_batch(name: str = "aspirin", n_frames: int = 256, cutoff: float = 5.0,
                   jitter: float = 0.05, seed: int = 0) -> Dict[str, torch.Tensor]:
    """``n_frames`` jittered copies of a molecule: R_b = R0 + jitter * N(0,1), zero cell,
    non-periodic, per-frame neighbour list (SURVEY.md §8(d) cfg 1-4)."""
    if name == "aspirin":
        Z0, R0 = ASPIRIN_Z, np.asarray(ASPIRIN_R)
    elif name == "ethanol":
        Z0, R0 = ETHANOL_Z, np.asarray(ETHANOL_R)
    else:
        raise ValueError(name)
    rng = np.random.RandomState(seed)
    systems = []
    for _ in range(n_frames):
        R = R0 + jitter * rng.randn(*R0.shape) if jitter > 0 else R0.copy()
        ii, jj = neighbor_pairs_open(R, cutoff)
        systems.append({"Z": Z0, "R": R, "idx_i": ii, "idx_j": jj})
    return collate(systems)


def blob_molecule_batch(n_atoms: int, n_frames: int, cutoff: float = 5.0, spacing: float = 1.9, jitter: float = 0.12,
                        seed: int = 0) -> Dict[str, torch.Tensor]:
    """``n_frames`` jittered copies of a compact synthetic molecule of ``n_atoms`` atoms (the ``n_atoms`` sites of a
    simple-cubic lattice of the given spacing nearest to its centre, H / C / O labels): stands in for the 29-atom QM9 and
    the 42-370-atom MD22 systems (datasets/qm9.py, datasets/md22.py) whose pair count per molecule exceeds what the
    molecule-resident kernels hold.  Non-periodic, full symmetric per-frame lists, like ``molecule_batch``."""
    rng = np.random.RandomState(seed)
    m = int(math.ceil(n_atoms ** (1.0 / 3.0))) + 2
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    g -= g.mean(0) + np.array([0.11, 0.07, 0.03])         # break the ties of the centred lattice
    order = np.argsort((g * g).sum(1), kind="stable")[:n_atoms]
    R0 = spacing * g[order]
    Z0 = rng.choice([1, 6, 8], size=n_atoms, p=[0.45, 0.4, 0.15]).tolist()
    systems = []
    for _ in range(n_frames):
        R = R0 + jitter * rng.randn(*R0.shape)
        ii, jj = neighbor_pairs_open(R, cutoff)
        systems.append({"Z": Z0, "R": R, "idx_i": ii, "idx_j": jj})
    return collate(systems)


def random_graph_batch(n_atoms: int, degree: int, seed: int = 0, box: float = 0.0,
                       sort: bool = True) -> Dict[str, torch.Tensor]:
    """Fixed-degree random directed graph (north_star's padded-neighbour sweep).  Not
    symmetric; exercises the general (non-symmetric) code path."""
    rng = np.random.RandomState(seed)
    idx_i = np.repeat(np.arange(n_atoms, dtype=np.int64), degree)
    idx_j = rng.randint(0, n_atoms, size=n_atoms * degree).astype(np.int64)
    clash = idx_j == idx_i
    idx_j[clash] = (idx_j[clash] + 1) % n_atoms
    if not sort:
        perm = rng.permutation(len(idx_i))
        idx_i, idx_j = idx_i[perm], idx_j[perm]
    # positions chosen so that distances spread over (0.8, 6.0): direct r_ij, no geometry
    r = rng.randn(n_atoms * degree, 3).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    r *= rng.uniform(0.8, 6.0, size=(n_atoms * degree, 1)).astype(np.float32)
    return {
        "Z": torch.from_numpy(rng.randint(1, 10, size=n_atoms).astype(np.int64)),
        "r_ij": torch.from_numpy(r),
        "idx_i": torch.from_numpy(idx_i),
        "idx_j": torch.from_numpy(idx_j),
        "idx_m": torch.zeros(n_atoms, dtype=torch.int64),
        "n_mol": 1,
    }


def ring_graph_batch(n_atoms: int, degree: int, seed: int = 0, dmin: float = 0.8, dmax: float = 4.9) -> Dict[str, torch.Tensor]:
    """Fixed-degree SYMMETRIC graph (north_star's padded-neighbour sweep): atom i is linked to i +- 1 .. i +- degree/2
    (mod n_atoms), both directions present, ``idx_i`` ascending and ``idx_j`` ascending within a row -- the structure of
    every reference neighbour list -- with antisymmetric pair vectors r_(j<-i) = -r_(i<-j) given directly (no positions)
    and lengths uniform in (dmin, dmax)."""
    assert degree % 2 == 0 and n_atoms > degree
    rng = np.random.RandomState(seed)
    half = degree // 2
    i0 = np.repeat(np.arange(n_atoms, dtype=np.int64), half)
    m = np.tile(np.arange(1, half + 1, dtype=np.int64), n_atoms)
    j0 = (i0 + m) % n_atoms
    v = rng.randn(n_atoms * half, 3).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v *= rng.uniform(dmin, dmax, size=(n_atoms * half, 1)).astype(np.float32)
    ii = np.concatenate([i0, j0])
    jj = np.concatenate([j0, i0])
    r = np.concatenate([v, -v])
    order = np.lexsort((jj, ii))
    return {
        "Z": torch.from_numpy(rng.randint(1, 10, size=n_atoms).astype(np.int64)),
        "r_ij": torch.from_numpy(np.ascontiguousarray(r[order])),
        "idx_i": torch.from_numpy(ii[order]),
        "idx_j": torch.from_numpy(jj[order]),
        "idx_m": torch.zeros(n_atoms, dtype=torch.int64),
        "n_mol": 1,
    }


def water_box(n_side: int = 22, cutoff: float = 5.0, seed: int = 0, jitter: float = 0.3):
    """Bulk-water-like periodic box (SURVEY.md §8(d) cfg 5): n_side^3 molecules on a jittered
    cubic lattice at 0.0334 molecules/A^3, rigid TIP3P-like geometry, random orientations.
    n_side=22 -> 10 648 molecules / 31 944 atoms.  Neighbour list by a host cell list; full
    symmetric list with cell offsets, sorted by i."""
    rng = np.random.RandomState(seed)
    n_mol = n_side ** 3
    a = (1.0 / 0.0334) ** (1.0 / 3.0)
    L = a * n_side
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centers = (g + 0.5) * a + jitter * rng.randn(n_mol, 3)
    # rigid water: O at origin, H at 0.9572 A, angle 104.52 deg
    th = math.radians(104.52) / 2
    mol = np.array([[0, 0, 0], [0.9572 * math.sin(th), 0.9572 * math.cos(th), 0],
                    [-0.9572 * math.sin(th), 0.9572 * math.cos(th), 0]])
    q = rng.randn(n_mol, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    Rm = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    R = (centers[:, None, :] + np.einsum("mij,aj->mai", Rm, mol)).reshape(-1, 3)
    R = np.mod(R, L).astype(np.float32)
    Z = np.tile(np.array([8, 1, 1], dtype=np.int64), n_mol)
    ii, jj, S = cell_list_pairs(R, L, cutoff)
    offsets = (S * np.float32(L)).astype(np.float32)
    return {
        "Z": torch.from_numpy(Z), "R": torch.from_numpy(R),
        "idx_i": torch.from_numpy(ii), "idx_j": torch.from_numpy(jj),
        "offsets": torch.from_numpy(offsets),
        "idx_m": torch.zeros(len(Z), dtype=torch.int64), "n_mol": 1,
        "cell": torch.eye(3) * L,
    }


def cell_list_pairs(R: np.ndarray, L: float, cutoff: float):
    """Host cell list for a cubic periodic box: all directed pairs with
    |R_j - R_i + S L| < cutoff (float32 test), sorted by (i, j).  Requires L >= 2 cutoff so
    that every pair has at most one image inside the cutoff."""
    assert L >= 2 * cutoff
    n = len(R)
    nc = max(1, int(math.floor(L / cutoff)))
    w = L / nc
    c = np.minimum((R / w).astype(np.int64), nc - 1)
    cid = (c[:, 0] * nc + c[:, 1]) * nc + c[:, 2]
    order = np.argsort(cid, kind="stable")
    cid_s = cid[order]
    starts = np.searchsorted(cid_s, np.arange(nc ** 3))
    ends = np.searchsorted(cid_s, np.arange(nc ** 3), side="right")
    out_i, out_j, out_S = [], [], []
    R32 = R.astype(np.float32)
    shifts = [(a, b, cc) for a in (-1, 0, 1) for b in (-1, 0, 1) for cc in (-1, 0, 1)]
    if nc < 3:
        # neighbouring cells alias; use unique cell offsets only
        shifts = list({(a % nc, b % nc, cc % nc): (a, b, cc) for a, b, cc in shifts}.values())
    cell_idx = np.arange(nc ** 3)
    cx, cy, cz = cell_idx // (nc * nc), (cell_idx // nc) % nc, cell_idx % nc
    for (a, b, cc) in shifts:
        nx, ny, nz = cx + a, cy + b, cz + cc
        nb = ((nx % nc) * nc + (ny % nc)) * nc + (nz % nc)
        for ca in range(nc ** 3):
            ia = order[starts[ca]:ends[ca]]
            ib = order[starts[nb[ca]]:ends[nb[ca]]]
            if len(ia) == 0 or len(ib) == 0:
                continue
            diff = R32[ib][None, :, :] - R32[ia][:, None, :]
            Sx = -np.round(diff / np.float32(L))
            diff = diff + Sx * np.float32(L)
            d = np.sqrt((diff * diff).sum(-1, dtype=np.float32))
            m = d < np.float32(cutoff)
            m &= ia[:, None] != ib[None, :]
            pi, pj = np.nonzero(m)
            out_i.append(ia[pi]); out_j.append(ib[pj]); out_S.append(Sx[pi, pj])
    ii = np.concatenate(out_i); jj = np.concatenate(out_j); S = np.concatenate(out_S)
    key = ii * n + jj
    o = np.argsort(key, kind="stable")
    return ii[o].astype(np.int64), jj[o].astype(np.int64), S[o].astype(np.float32)
