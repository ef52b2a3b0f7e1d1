"""Neighbour lists on the device.

* :func:`neighbor_list` -- the batched primitive over ``spk_nbl_count_f32`` / ``spk_nbl_fill_f32``.
* :class:`HipNeighborList` -- mirror of the reference's neighbour-list plug-in interface
  (``NeighborListTransform._build_neighbor_list(Z, positions, cell, pbc, cutoff)``,
  transform/neighborlist.py:159-211); drop-in for ``TorchNeighborList`` / ``ASENeighborList`` /
  ``MatScipyNeighborList``.
* :class:`NeighborListMD` -- mirror of md/neighborlist_md.py:12-189 (cutoff shell, rebuild when an atom
  moved more than half the shell, buffer-zone filter), but ONE batched device build for all
  replicas / molecules instead of a ``.cpu()`` round trip and a Python loop per molecule (:126-159,
  :219-229).

No CPU path: the search runs in ``libspk_hip.so`` or raises.
"""
import ctypes
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, properties
from ._lib import SpkHipError, check, fptr, iptr, lib, stream

__all__ = ["neighbor_list", "NeighborListTransform", "HipNeighborList", "NeighborListMD"]


def neighbor_list(R: torch.Tensor, cutoff: float, idx_m: Optional[torch.Tensor] = None,
                  cell: Optional[torch.Tensor] = None, pbc: Optional[torch.Tensor] = None,
                  n_systems: Optional[int] = None, return_shifts: bool = False) -> Dict[str, torch.Tensor]:
    """All directed pairs (i, j, S) with ``|R_j - R_i + S.cell| < cutoff`` for a batch of systems.

    R [N,3] float32 (ROCm device); idx_m [N] int64 ascending system index (``_idx_m``) or None for
    one system; cell [M,3,3] (or [3,3]); pbc [M,3] / [3M] / [3] bool.  Returns ``_idx_i``, ``_idx_j``
    (int64, idx_i ascending), ``_offsets`` [E,3] = S.cell, ``rowptr`` [N+1] int32 (CSR of idx_i) and,
    on request, ``shifts`` [E,3] int32.
    """
    _lib.require_device(R)
    if R.dtype != torch.float32:
        raise SpkHipError("neighbor_list: positions must be float32, got %s" % R.dtype)
    Rc = R.detach().contiguous()
    N = int(Rc.shape[0])
    dev = Rc.device
    if idx_m is not None:
        idx_m = idx_m.to(device=dev, dtype=torch.int64).contiguous()
        if n_systems is None:
            n_systems = int(idx_m[-1]) + 1 if N > 0 else 1
    else:
        n_systems = 1
    M = int(n_systems)
    cellc = pbcc = None
    if pbc is not None:
        pbcc = pbc.to(device=dev).reshape(-1, 3).to(torch.bool).contiguous()
        if pbcc.shape[0] != M:
            raise SpkHipError("neighbor_list: pbc describes %d systems, expected %d" % (pbcc.shape[0], M))
    if cell is not None:
        cellc = cell.to(device=dev, dtype=torch.float32).reshape(-1, 3, 3).contiguous()
        if cellc.shape[0] != M:
            raise SpkHipError("neighbor_list: cell describes %d systems, expected %d" % (cellc.shape[0], M))
    L = lib()
    ws = torch.empty(int(L.spk_nbl_workspace_bytes(N, M)), dtype=torch.uint8, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    n_edges = ctypes.c_int64(0)
    pbc_ptr = ctypes.c_void_p(pbcc.data_ptr()) if pbcc is not None else None
    ws_ptr = ctypes.c_void_p(ws.data_ptr())
    with torch.cuda.device(dev):
        check(L.spk_nbl_count_f32(fptr(Rc), iptr(idx_m) if idx_m is not None else None, fptr(cellc), pbc_ptr, N, M,
                                  float(cutoff), ws_ptr, iptr(rowptr, torch.int32), ctypes.byref(n_edges), stream()))
        E = int(n_edges.value)
        idx_i = torch.empty(E, dtype=torch.int64, device=dev)
        idx_j = torch.empty(E, dtype=torch.int64, device=dev)
        offsets = torch.empty((E, 3), dtype=torch.float32, device=dev)
        shifts = torch.empty((E, 3), dtype=torch.int32, device=dev) if return_shifts else None
        check(L.spk_nbl_fill_f32(fptr(Rc), iptr(idx_m) if idx_m is not None else None, N, M, float(cutoff), ws_ptr,
                                 iptr(rowptr, torch.int32), E, iptr(idx_i), iptr(idx_j),
                                 iptr(shifts, torch.int32) if shifts is not None else None, fptr(offsets), stream()))
    out = {properties.idx_i: idx_i, properties.idx_j: idx_j, properties.offsets: offsets, "rowptr": rowptr}
    if return_shifts:
        out["shifts"] = shifts
    return out


class NeighborListTransform(nn.Module):
    """Base class of neighbour-list transforms, same contract as the reference
    (transform/neighborlist.py:159-211): ``forward`` reads Z / R / cell / pbc of ONE system from the
    dict and writes ``_idx_i``, ``_idx_j``, ``_offsets``."""

    is_preprocessor: bool = True
    is_postprocessor: bool = False

    def __init__(self, cutoff: float):
        super().__init__()
        self._cutoff = cutoff

    def datamodule(self, value):
        pass

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        Z = inputs[properties.Z]
        R = inputs[properties.R]
        cell = inputs[properties.cell].view(3, 3)
        pbc = inputs[properties.pbc]
        idx_i, idx_j, offset = self._build_neighbor_list(Z, R, cell, pbc, self._cutoff)
        inputs[properties.idx_i] = idx_i.detach()
        inputs[properties.idx_j] = idx_j.detach()
        inputs[properties.offsets] = offset
        return inputs

    def _build_neighbor_list(self, Z, positions, cell, pbc, cutoff):
        raise NotImplementedError


class HipNeighborList(NeighborListTransform):
    """Cell-list search on the GPU.  Tensors may live anywhere (the data pipeline hands over CPU
    tensors, often float64): the search runs in float32 on the current ROCm device, indices come
    back on the device of ``positions`` and ``offsets = S @ cell`` is evaluated in the dtype of the
    positions exactly like the reference back-ends do (transform/neighborlist.py:225, :455-457)."""

    def __init__(self, cutoff: float, device: Optional[str] = None):
        super().__init__(cutoff)
        self._device = device

    def _build_neighbor_list(self, Z, positions, cell, pbc, cutoff):
        src = positions.device
        dev = torch.device(self._device) if self._device is not None else (src if src.type == "cuda" else torch.device("cuda", torch.cuda.current_device()))
        R32 = positions.detach().to(device=dev, dtype=torch.float32)
        nl = neighbor_list(R32, float(cutoff), None, cell.detach().reshape(1, 3, 3), pbc.reshape(1, 3), return_shifts=True)
        S = nl["shifts"].to(device=src, dtype=positions.dtype)
        offset = torch.mm(S, cell.to(device=src, dtype=positions.dtype))
        return nl[properties.idx_i].to(src), nl[properties.idx_j].to(src), offset


class NeighborListMD:
    """Neighbour list for molecular dynamics: all replicas / molecules of the batch in one device
    build with ``cutoff + cutoff_shell``; rebuilt when any atom moved further than half the shell or a
    cell changed (md/neighborlist_md.py:55-98); pairs in the buffer zone are filtered out per call
    (``d <= cutoff``, :161-189) unless ``filter_buffer=False``, which keeps the list -- and therefore
    every tensor shape of the force call -- unchanged between rebuilds so that the force call can be
    replayed as a HIP graph (the cosine cutoff makes the extra pairs contribute exactly zero).

    ``base_nbl`` / ``collate_fn`` are accepted for signature compatibility and ignored."""

    def __init__(self, cutoff: float, cutoff_shell: float, base_nbl=None, requires_triples: bool = False,
                 collate_fn=None, filter_buffer: bool = True):
        if requires_triples:
            raise NotImplementedError("atom triples are outside the SchNet / PaiNN hot path")
        self.cutoff = cutoff
        self.cutoff_shell = cutoff_shell
        self.cutoff_full = cutoff + cutoff_shell
        self.requires_triples = requires_triples
        self.filter_buffer = filter_buffer
        self.previous_positions = None
        self.previous_cells = None
        self._list = None
        self.n_builds = 0

    def _update_required(self, positions, cells) -> bool:
        if self._list is None or self.previous_positions is None or self.previous_positions.shape != positions.shape:
            return True
        moved = ((self.previous_positions - positions) ** 2).sum(1).max() > (0.5 * self.cutoff_shell) ** 2
        if cells is not None and self.previous_cells is not None:
            moved = moved | torch.any(self.previous_cells != cells)
        return bool(moved)   # one scalar D2H per step

    def get_neighbors(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        positions = inputs[properties.R]
        idx_m = inputs[properties.idx_m]
        cells = inputs.get(properties.cell)
        pbc = inputs.get(properties.pbc)
        n_molecules = int(inputs[properties.n_atoms].shape[0])
        if self._update_required(positions, cells):
            self.previous_positions = positions.detach().clone()
            self.previous_cells = cells.detach().clone() if cells is not None else None
            self._list = neighbor_list(positions.detach().float(), self.cutoff_full, idx_m, cells, pbc, n_systems=n_molecules)
            self.n_builds += 1
        nl = self._list
        out = {properties.idx_i: nl[properties.idx_i], properties.idx_j: nl[properties.idx_j],
               properties.offsets: nl[properties.offsets].to(positions.dtype)}
        if self.filter_buffer:
            Rij = positions[out[properties.idx_j]] - positions[out[properties.idx_i]] + out[properties.offsets]
            keep = torch.linalg.norm(Rij, dim=1) <= self.cutoff
            out = {k: v[keep] for k, v in out.items()}
        return out
