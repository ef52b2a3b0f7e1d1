"""Collate-side wire format (second half): mirror of ``schnetpack.data.loader``
(data/loader.py:13-86) whose collate function -- running in the DataLoader WORKERS -- also produces what the device
kernels otherwise derive per neighbour list with ``spk_edge_plan`` and its host round trips:

* CSR row pointers of ``_idx_i`` (int32), the reverse-edge map, the canonical edge of every undirected pair, the pair of
  every directed edge (all int32),
* the block-diagonal grouping of the batch (molecules merged into groups of <= 32 atoms: the work units of the
  molecule-resident kernels, ``spk_schnet_mol.hip``),
* flags: sorted / symmetric, and -- when the model cutoff is given -- whether the list carries pairs beyond it (skin lists).

``install_plan(inputs)`` hands these arrays to the operator library's plan cache after the batch has been moved to the
device, so the first force call on a new batch launches no plan kernels and performs no device-to-host copy.
Batches without the extra keys (the reference's own collate function) keep working: the plan is then derived on the device.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

from . import properties as structure

__all__ = ["AtomsLoader", "atoms_collate_fn", "WireCollate", "install_plan", "host_plan"]

PLAN_KEYS = ("_spk_rowptr", "_spk_rev", "_spk_half", "_spk_edge_pair", "_spk_grp_atom0", "_spk_grp_pair0", "_spk_plan_meta")
MAX_GROUP_ATOMS = 32


def atoms_collate_fn(batch: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Same result as the reference's ``_atoms_collate_fn`` (data/loader.py:13-58): per-atom / per-system tensors are
    concatenated, index tensors are kept once as ``<key>_local`` and once shifted by the atom offset of their system,
    ``_idx_m`` is the system index of every atom, triple indices are shifted by the pair offset."""
    elem = batch[0]
    pair_keys = (structure.idx_i, structure.idx_j, structure.idx_i_triples)
    triple_keys = (structure.idx_j_triples, structure.idx_k_triples)
    out: Dict[str, torch.Tensor] = {}
    for key in elem:
        cat = torch.cat([d[key] for d in batch], 0)
        if key in pair_keys:
            out[key + "_local"] = cat
        elif key not in triple_keys:
            out[key] = cat
    n_atoms = out[structure.n_atoms]
    start = torch.cumsum(n_atoms, 0) - n_atoms                      # first atom of every system
    out[structure.idx_m] = torch.repeat_interleave(torch.arange(len(batch)), n_atoms, dim=0)
    for key in pair_keys:
        if key in elem:
            counts = torch.tensor([int(d[key].shape[0]) for d in batch])
            out[key] = out[key + "_local"] + torch.repeat_interleave(start, counts)
    for key in triple_keys:
        if key in elem:
            n_pairs = torch.tensor([int(d[structure.idx_j].shape[0]) for d in batch])
            pstart = torch.cumsum(n_pairs, 0) - n_pairs
            counts = torch.tensor([int(d[key].shape[0]) for d in batch])
            out[key] = torch.cat([d[key] for d in batch], 0) + torch.repeat_interleave(pstart, counts)
    return out


def host_plan(idx_i: np.ndarray, idx_j: np.ndarray, offsets: Optional[np.ndarray], n_atoms: int, distances: Optional[np.ndarray] = None,
              cutoff: Optional[float] = None) -> Dict[str, np.ndarray]:
    """What ``spk_edge_plan`` + the plan-time grouping derive on the device, on the host (numpy; E log E).

    meta = [sorted, symmetric, n_half, n_groups, max_group_atoms, max_group_pairs, filter_pairs (-1 unknown), n_tiles_grouped]."""
    E = int(idx_i.shape[0])
    i64, j64 = idx_i.astype(np.int64), idx_j.astype(np.int64)
    is_sorted = bool(E == 0 or np.all(i64[1:] >= i64[:-1]))
    in_range = bool(E == 0 or (i64.min() >= 0 and j64.min() >= 0 and i64.max() < n_atoms and j64.max() < n_atoms))
    if not in_range:
        raise ValueError("host_plan: neighbour index out of range [0, %d)" % n_atoms)
    rowptr = np.zeros(n_atoms + 1, dtype=np.int32)
    meta = np.array([int(is_sorted), 0, 0, 0, 0, 0, -1, 0], dtype=np.int64)
    empty = np.zeros(0, dtype=np.int32)
    res = {"rowptr": rowptr, "rev": np.full(max(E, 1), -1, dtype=np.int32), "half": empty, "edge_pair": empty,
           "grp_atom0": empty, "grp_pair0": empty, "meta": meta}
    if not is_sorted:
        return res
    np.cumsum(np.bincount(i64, minlength=n_atoms), out=rowptr[1:])
    if E == 0:
        meta[1] = 1
        return res
    # reverse edge: (i, j, o) <-> (j, i, -o), offsets compared bit for bit (the reversed image shift is the exact negation)
    off = np.zeros((E, 3), dtype=np.float32) if offsets is None else np.ascontiguousarray(offsets, dtype=np.float32)
    ob = (off + np.float32(0.0)).view(np.int32)            # + 0.0: -0.0 and 0.0 compare equal
    nb = (-off + np.float32(0.0)).view(np.int32)
    fwd = np.lexsort((ob[:, 2], ob[:, 1], ob[:, 0], j64, i64))
    bwd = np.lexsort((nb[:, 2], nb[:, 1], nb[:, 0], i64, j64))
    same = (np.array_equal(i64[fwd], j64[bwd]) and np.array_equal(j64[fwd], i64[bwd]) and np.array_equal(ob[fwd], nb[bwd]))
    if same:       # duplicates of one (i, j, offset) would make the pairing ambiguous: the device plan rejects them too
        kf = np.stack([i64[fwd], j64[fwd], ob[fwd, 0], ob[fwd, 1], ob[fwd, 2]], 1)
        same = not bool(np.any(np.all(kf[1:] == kf[:-1], axis=1)))
    if not same:
        return res
    rev = np.empty(E, dtype=np.int32)
    rev[fwd] = bwd.astype(np.int32)
    if np.any(rev == np.arange(E)):                           # an edge that is its own reverse (i == j, zero shift)
        return res
    half = np.nonzero(rev > np.arange(E))[0].astype(np.int32)
    if 2 * half.shape[0] != E:
        return res
    edge_pair = np.empty(E, dtype=np.int32)
    k = np.arange(half.shape[0], dtype=np.int32)
    edge_pair[half] = k
    edge_pair[rev[half]] = k
    res.update(rev=rev, half=half, edge_pair=edge_pair)
    meta[1], meta[2] = 1, half.shape[0]
    if distances is not None and cutoff is not None:
        meta[6] = int(float(np.mean(distances >= cutoff)) > 0.05)
    # block-diagonal groups: connected atom ranges that no edge leaves, merged greedily up to MAX_GROUP_ATOMS atoms
    mj = np.arange(n_atoms, dtype=np.int64)
    np.maximum.at(mj, i64, j64)
    ends = np.nonzero(np.maximum.accumulate(mj) == np.arange(n_atoms))[0] + 1
    sizes = np.diff(np.concatenate([[0], ends]))
    if sizes.size == 0 or sizes.max() > MAX_GROUP_ATOMS:
        return res
    atom0 = [0]
    cur = 0
    for sz in sizes.tolist():
        if cur + sz > MAX_GROUP_ATOMS and cur > 0:
            atom0.append(atom0[-1] + cur)
            cur = 0
        cur += sz
    atom0.append(atom0[-1] + cur)
    atom0 = np.asarray(atom0, dtype=np.int64)
    pair0 = np.searchsorted(i64[half], atom0).astype(np.int32)
    tiles = (np.diff(pair0.astype(np.int64)) + 31) // 32
    res.update(grp_atom0=atom0.astype(np.int32), grp_pair0=pair0)
    meta[3], meta[4], meta[5], meta[7] = atom0.shape[0] - 1, int(np.diff(atom0).max()), int(np.diff(pair0).max()), int(tiles.sum())
    return res


class WireCollate:
    """Collate function for DataLoader workers: ``atoms_collate_fn`` + the host plan of the collated neighbour list
    (keys ``_spk_*``).  ``model_cutoff``: when the list was built with a larger radius (skin), also decide on the host whether
    the per-call pair compaction should run."""

    def __init__(self, model_cutoff: Optional[float] = None):
        self.model_cutoff = model_cutoff

    def __call__(self, batch):
        out = atoms_collate_fn(batch)
        if structure.idx_i not in out:
            return out
        ii, jj = out[structure.idx_i].numpy(), out[structure.idx_j].numpy()
        off = out[structure.offsets].numpy() if structure.offsets in out else None
        dist = None
        if self.model_cutoff is not None and structure.R in out:
            R = out[structure.R].numpy().astype(np.float32)
            r = R[jj] - R[ii] + (off.astype(np.float32) if off is not None else 0.0)
            dist = np.sqrt((r * r).sum(1))
        plan = host_plan(ii, jj, off, int(out[structure.Z].shape[0]), dist, self.model_cutoff)
        for key, name in zip(PLAN_KEYS, ("rowptr", "rev", "half", "edge_pair", "grp_atom0", "grp_pair0", "meta")):
            out[key] = torch.from_numpy(plan[name])
        return out


def to_device(batch: Dict[str, torch.Tensor], device, non_blocking: bool = True) -> Dict[str, torch.Tensor]:
    """Move a collated batch to ``device``.  ``_spk_plan_meta`` (eight scalars that the HOST reads when the plan is installed)
    stays where the collate function put it: on the device it would cost a blocking device-to-host copy per batch."""
    return {k: (v if k == PLAN_KEYS[6] else v.to(device, non_blocking=non_blocking)) for k, v in batch.items()}


def _validate_plan(inputs, meta):
    """Debug check of an installed plan (synchronises): the library only checks the SIZES of the arrays it is handed."""
    E = int(inputs[structure.idx_i].shape[0])
    N = int(inputs[structure.Z].shape[0])
    rowptr, rev, half, edge_pair, ga, gp = (inputs[k] for k in PLAN_KEYS[:6])

    def in_range(t, lo, hi, what):
        if t.numel() and (int(t.min()) < lo or int(t.max()) >= hi):
            raise ValueError("install_plan: %s outside [%d, %d)" % (what, lo, hi))

    in_range(rowptr, 0, E + 1, "rowptr")
    if bool((rowptr[1:] < rowptr[:-1]).any()) or int(rowptr[-1]) != E:
        raise ValueError("install_plan: rowptr is not a CSR row pointer of the list")
    if meta[1] and E:
        in_range(rev[:E], 0, E, "rev")
        in_range(half, 0, E, "half")
        in_range(edge_pair, 0, meta[2], "edge_pair")
        if meta[3] > 0:
            in_range(ga, 0, N + 1, "grp_atom0")
            in_range(gp, 0, meta[2] + 1, "grp_pair0")


def install_plan(inputs: Dict[str, torch.Tensor], validate: bool = False) -> bool:
    """Hand the host-made plan of a batch (arrays already on the device) to the operator library: the plan cache entry of
    ``(inputs["_idx_i"], inputs["_idx_j"])`` is created from the ``_spk_*`` tensors without a kernel launch or a sync.
    ``_spk_plan_meta`` must still be a HOST tensor (move batches with :func:`to_device`): the host reads its eight scalars here,
    and from the device that would be a blocking copy per batch -- refused instead of paid silently.  ``validate=True`` range-checks
    the arrays first (debug; synchronises).  Returns False when the batch carries no plan (then the device derives it on first use)."""
    if PLAN_KEYS[0] not in inputs:
        return False
    from . import torchops  # noqa: F401
    meta = inputs[PLAN_KEYS[6]]
    if meta.device.type != "cpu":
        raise ValueError("install_plan: '_spk_plan_meta' is on %s; keep it on the host (schnetpack_amd.data.to_device moves "
                         "everything else) -- reading it back would synchronise every batch" % meta.device)
    meta = [int(v) for v in meta.tolist()]
    if validate:
        _validate_plan(inputs, meta)
    torch.ops.spk_hip.edge_plan_install(inputs[structure.idx_i], inputs[structure.idx_j], int(inputs[structure.Z].shape[0]),
                                        inputs[PLAN_KEYS[0]], inputs[PLAN_KEYS[1]], inputs[PLAN_KEYS[2]], inputs[PLAN_KEYS[3]],
                                        inputs[PLAN_KEYS[4]], inputs[PLAN_KEYS[5]], meta)
    return True


class AtomsLoader(DataLoader):
    """Mirror of ``schnetpack.data.AtomsLoader`` (data/loader.py:61-86); ``collate_fn`` defaults to the wire-format collate."""

    def __init__(self, dataset: Dataset, batch_size: Optional[int] = 1, shuffle: bool = False, sampler: Optional[Sampler] = None,
                 batch_sampler: Optional[Sampler[Sequence[int]]] = None, num_workers: int = 0, collate_fn=None,
                 pin_memory: bool = False, **kwargs):
        super().__init__(dataset=dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, batch_sampler=batch_sampler,
                         num_workers=num_workers, collate_fn=collate_fn if collate_fn is not None else WireCollate(),
                         pin_memory=pin_memory, **kwargs)
