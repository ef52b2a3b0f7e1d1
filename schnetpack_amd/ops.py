"""``ctypes`` face of the C ABI (include/spk_hip.h) for callers and tests that want the RAW launchers: thin wrappers that
hand device pointers of torch tensors to ``libspk_hip.so`` -- ``_scatter_raw`` / ``_gather_raw`` (``spk_scatter_add_f32`` /
``spk_gather_f32``), ``dense_raw`` (``spk_dense_f32``), ``radial_struct`` (``spk_radial_t``), and :class:`EdgePlan`, a
``spk_graph_t`` view of the neighbour-list plan.

There is ONE implementation of the plan logic and ONE autograd layer, both in the operator library (``csrc/spk_torch.cpp``:
``get_plan`` / ``build_groups``, the C++ ``torch::autograd::Function``s).  This module owns neither: :class:`EdgePlan` wraps the
arrays of the library's plan (``torch.ops.spk_hip.edge_plan_arrays``), and the differentiable entry points kept here for
convenience (``scatter_add``, ``gather``, ``pairwise_vectors``, ``dense``) ARE the product operators ``torch.ops.spk_hip.*``.
(Until round 3 this file carried a second, Python, implementation of both -- plan derivation and ``autograd.Function``s --
that the package itself no longer used.)

There is no CPU path: CPU tensors raise ``SpkHipError``.
"""
import ctypes

import torch

from . import _lib
from ._lib import SpkHipError, check, fptr, iptr, lib, stream
from .torchops import ops as _T


def _loud(fn, *args):
    """Product operators raise RuntimeError (TORCH_CHECK / the C ABI's status); the raw face keeps its own exception type."""
    try:
        return fn(*args)
    except SpkHipError:
        raise
    except RuntimeError as exc:
        raise SpkHipError(str(exc)) from exc


# ----------------------------------------------------------------------------- plans
class EdgePlan:
    """``spk_graph_t`` of one neighbour list: CSR row pointers, sorted / symmetric flags, reverse-edge map, canonical pairs,
    ``edge_pair``, <= 32-atom groups -- the arrays of the operator library's plan (one derivation: ``spk_edge_plan`` + the
    plan-time grouping in ``spk_torch.cpp``), wrapped for raw C-ABI calls.  Holds references to the index tensors."""

    def __init__(self, idx_i, idx_j, n_atoms, r_ij=None, want_groups=None):
        _lib.require_device(idx_i, idx_j)
        self._src = (idx_i, idx_j)
        self.n_atoms = int(n_atoms)
        arrs = _loud(_T.edge_plan_arrays, idx_i, idx_j, self.n_atoms, r_ij.detach().float().contiguous() if r_ij is not None else None)
        rowptr, rev, half, edge_pair, g_atom0, g_pair0, g_tile0, meta, self.idx_i, self.idx_j = arrs
        m = [int(v) for v in meta.tolist()]
        self.n_edges = int(self.idx_i.shape[0])
        self.rowptr, self.rev = rowptr, rev
        self.sorted, self.symmetric = bool(m[0]), bool(m[1])
        n_half, n_groups, max_ga, max_gp, n_tiles_g = m[2], m[3], m[4], m[5], m[7]
        self.half = half if (self.symmetric and n_half > 0) else None
        self.edge_pair = edge_pair if (self.symmetric and n_half > 0 and edge_pair.numel() > 0) else None
        self.groups = (g_atom0, g_pair0, g_tile0, max_ga, n_tiles_g, max_gp) if (n_groups > 0 and want_groups is not False) else None
        gp = self.groups
        self._graph = _lib.GraphT(self.n_atoms, self.n_edges, iptr(self.idx_i), iptr(self.idx_j),
                                  iptr(self.rowptr, torch.int32) if self.sorted else None,
                                  int(self.sorted), int(self.symmetric),
                                  iptr(self.rev, torch.int32) if self.symmetric else None,
                                  iptr(self.half, torch.int32) if self.half is not None else None, n_half if self.half is not None else 0,
                                  iptr(gp[0], torch.int32) if gp else None, iptr(gp[1], torch.int32) if gp else None,
                                  iptr(gp[2], torch.int32) if gp else None, n_groups if gp else 0, max_ga if gp else 0,
                                  n_tiles_g if gp else 0, 0, 0, None,
                                  iptr(self.edge_pair, torch.int32) if self.edge_pair is not None else None,
                                  max_gp if gp else 0, 0, None, None)
        self.blocks = None
        self.transposed = None

    def graph(self):
        return ctypes.byref(self._graph)

    def build_transposed(self):
        """The list sorted by neighbour (``spk_transposed_build``), attached to the graph: asymmetric sorted lists then run their
        transposed sums as row passes.  No host synchronisation."""
        dev = self.idx_i.device
        E, N = self.n_edges, self.n_atoms
        bufs = dict(idx_i=torch.empty(max(E, 1), dtype=torch.int64, device=dev), idx_j=torch.empty(max(E, 1), dtype=torch.int64, device=dev),
                    rowptr=torch.zeros(N + 2, dtype=torch.int32, device=dev), perm=torch.empty(max(E, 1), dtype=torch.int32, device=dev),
                    r_perm=torch.empty(max(E, 1), 3, dtype=torch.float32, device=dev),
                    tmp=torch.empty(max(int(lib().spk_transpose_plan_bytes(E, N)), 16), dtype=torch.uint8, device=dev))
        with torch.cuda.device(dev):
            check(lib().spk_transposed_build(iptr(self.idx_i), iptr(self.idx_j), E, N, bufs["idx_i"].data_ptr(), bufs["idx_j"].data_ptr(),
                                             bufs["rowptr"].data_ptr(), bufs["perm"].data_ptr(), bufs["tmp"].data_ptr(), stream()))
        t = _lib.TransposedT(bufs["idx_i"].data_ptr(), bufs["idx_j"].data_ptr(), bufs["rowptr"].data_ptr(), bufs["perm"].data_ptr(), bufs["r_perm"].data_ptr())
        self.transposed, self._transposed_bufs = t, bufs
        self._graph.transposed = ctypes.addressof(t)
        return t

    def build_blocks(self, n_rbf=20, n_atom_basis=128, cap=0):
        """Block plan of the list (``spk_blocks_build``; include/spk_hip.h) for the box-regime PaiNN message kernels, attached to
        the graph.  Returns (usable, largest unique-neighbour count, tiles).  One device-to-host copy -- per list, not per call."""
        assert self.sorted, "block plans need a list sorted by idx_i"
        dev = self.idx_i.device
        sizes = (ctypes.c_int64 * 12)()
        check(lib().spk_blocks_sizes(self.n_atoms, self.n_edges, int(n_rbf), int(n_atom_basis), sizes))
        i32 = lambda n: torch.zeros(max(int(n), 1), dtype=torch.int32, device=dev)
        f32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.float32, device=dev)
        bufs = dict(sub_n=i32(sizes[0]), sub_u=i32(sizes[1]), uniq=i32(sizes[2]), jl=torch.zeros(max(int(sizes[3]), 1), dtype=torch.int16, device=dev),
                    atom_tile0=i32(sizes[4]), tile_info=i32(sizes[5]), blk_desc=i32(sizes[10]), stats=i32(4))
        b = _lib.BlocksT()
        for name in ("sub_n", "sub_u", "uniq", "jl", "atom_tile0", "tile_info", "blk_desc"):
            setattr(b, name, bufs[name].data_ptr())
        host = (ctypes.c_int32 * 4)()
        with torch.cuda.device(dev):
            check(lib().spk_blocks_build(self.graph(), int(n_rbf), int(cap), ctypes.byref(b), bufs["stats"].data_ptr(), host, stream()))
        ks, nt = int(sizes[9]), max(int(b.n_tiles), 1)
        bufs.update(apack=f32(nt * ks * 64), adpack=f32(nt * ks * 64), rec=f32(nt * 6 * 16), part=f32(sizes[8]))
        for name in ("apack", "adpack", "rec", "part"):
            setattr(b, name, bufs[name].data_ptr())
        self.blocks, self._block_bufs = b, bufs
        self._graph.blocks = ctypes.addressof(b) if b.ok else None
        return bool(b.ok), int(b.max_unique), int(b.n_tiles)

    # lists with pairs at or beyond the cutoff (MD skin lists): per-call compaction in the fused SchNet path
    filter_pairs = None     # None: undecided, see decide_filter()

    def set_filter(self, on: bool):
        self.filter_pairs = bool(on)
        self._graph.filter_pairs = 1 if on else 0

    def decide_filter(self, r_ij, cutoff, threshold=0.05):
        """Enable the per-call pair compaction if more than ``threshold`` of the pairs of the list are at or
        beyond the cutoff right now (one D2H sync, once per list)."""
        if self.filter_pairs is None:
            if self.n_edges == 0 or not self.symmetric:
                self.set_filter(False)
            else:
                d = torch.linalg.norm(r_ij.detach(), dim=1)
                self.set_filter(float((d >= cutoff).float().mean()) > threshold)
        return self.filter_pairs


def edge_plan(idx_i, idx_j, n_atoms, r_ij=None):
    """Plan of a neighbour list (the operator library caches it, keyed by the identity / version of the index tensors)."""
    return EdgePlan(idx_i, idx_j, n_atoms, r_ij)


def segment_rowptr(idx, dim_size):
    """CSR row pointers (int32, device) if ``idx`` is ascending, else None."""
    _lib.require_device(idx)
    if int(idx.shape[0]) == 0:
        return None
    plan = EdgePlan(idx, idx, dim_size, None)
    return plan.rowptr if plan.sorted else None


# ----------------------------------------------------------------------------- raw launchers
def _as_3d(x, dim):
    dim = dim % x.dim()
    outer = 1
    for s in x.shape[:dim]:
        outer *= int(s)
    inner = 1
    for s in x.shape[dim + 1:]:
        inner *= int(s)
    return dim, outer, int(x.shape[dim]), inner


def _scatter_raw(x, idx, dim_size, dim, rowptr):
    """``spk_scatter_add_f32``: segmented sum when ``rowptr`` is given (ascending index), float atomics otherwise."""
    x = x.contiguous()
    dim, outer, E, inner = _as_3d(x, dim)
    shape = list(x.shape)
    shape[dim] = int(dim_size)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().spk_scatter_add_f32(fptr(x), iptr(idx), iptr(rowptr, torch.int32) if rowptr is not None else None,
                                        outer, E, inner, int(dim_size), fptr(y), stream()))
    return y


def _gather_raw(x, idx, dim):
    x = x.contiguous()
    dim, outer, R, inner = _as_3d(x, dim)
    shape = list(x.shape)
    shape[dim] = int(idx.shape[0])
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().spk_gather_f32(fptr(x), iptr(idx), outer, R, int(idx.shape[0]), inner, fptr(y), stream()))
    return y


def radial_struct(kind, n_rbf, p0, p1, cutoff):
    return _lib.RadialT(int(kind), int(n_rbf), fptr(p0), fptr(p1) if p1 is not None else None, float(cutoff))


_ACT_IDS = {None: _lib.SPK_ACT_NONE, "none": _lib.SPK_ACT_NONE, "ssp": _lib.SPK_ACT_SSP, "silu": _lib.SPK_ACT_SILU}


def dense_raw(x, w, b, act, res=None, want_pre=False):
    """y = act(x w^T + b) (+ res) through ``spk_dense_f32``; x: [..., k]."""
    k = int(x.shape[-1])
    x2 = x.contiguous().view(-1, k)
    m, n_out = int(x2.shape[0]), int(w.shape[0])
    y = torch.empty((m, n_out), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if want_pre else None
    r2 = res.contiguous().view(-1, n_out) if res is not None else None
    with torch.cuda.device(x.device):
        check(lib().spk_dense_f32(fptr(x2), fptr(w.contiguous()), fptr(b.contiguous()) if b is not None else None,
                                  fptr(r2), fptr(y), fptr(pre), m, k, n_out, int(act), stream()))
    out_shape = tuple(x.shape[:-1]) + (n_out,)
    return y.view(out_shape), (pre.view(out_shape) if pre is not None else None)


def atomwise_supported(n_in, n_hidden, act):
    return bool(lib().spk_atomwise_supported(int(n_in), int(n_hidden), int(act)))


# ----------------------------------------------------------------------------- differentiable entry points = the product operators
def scatter_add(x, idx_i, dim_size, dim=0):
    _lib.require_device(x, idx_i)
    return _loud(_T.scatter_add, x, idx_i, int(dim_size), int(dim))


def gather(x, idx, dim=0, rowptr=None):
    _lib.require_device(x, idx)
    return _loud(_T.gather, x, idx, int(dim))


def pairwise_vectors(R, idx_i, idx_j, offsets=None):
    _lib.require_device(R, idx_i, idx_j)
    return _loud(_T.pairwise, R, idx_i, idx_j, offsets)


def dense(x, w, b=None, act=None, training=True):
    """``act(x w^T + b)``: the product operator (one autograd node for eval and training, csrc/spk_torch.cpp)."""
    _lib.require_device(x, w)
    a = _ACT_IDS[act] if not isinstance(act, int) else act
    return _loud(_T.dense, x, w, b, int(a))
