"""Python face of the HIP kernels: neighbour-list plans, autograd functions.

Two regimes:

* eval / MD / ASE (``module.training == False``): whole-representation fused functions
  (``SchNetFn`` / ``PaiNNFn``) -- one C call forward, one C call for the first-order backward
  w.r.t. geometry.  Not differentiable twice.
* training (force loss => double backward): primitives that are closed under differentiation --
  ``scatter_add`` <-> ``gather`` are each other's transposes (both HIP), Dense forward runs on
  the MFMA kernel with a backward written in differentiable torch algebra.

There is no CPU path: CPU tensors raise ``SpkHipError``.
"""
import collections
import ctypes

import torch

from . import _lib
from ._lib import SpkHipError, check, fptr, iptr, lib, stream


# ----------------------------------------------------------------------------- plans
class EdgePlan:
    """CSR row pointers + flags of one neighbour list (``spk_edge_plan``).  Built once per
    list (one 16-byte D2H sync) and cached; holds references to the index tensors."""

    def __init__(self, idx_i, idx_j, n_atoms, r_ij=None, want_groups=None):
        _lib.require_device(idx_i, idx_j)
        if want_groups is None:
            want_groups = True
        self.idx_i = idx_i.long().contiguous()
        self.idx_j = idx_j.long().contiguous()
        self.n_atoms = int(n_atoms)
        self.n_edges = int(self.idx_i.shape[0])
        dev = self.idx_i.device
        self.rowptr = torch.empty(self.n_atoms + 1, dtype=torch.int32, device=dev)
        self.rev = torch.full((max(self.n_edges, 1),), -1, dtype=torch.int32, device=dev)
        self.half = None
        scratch = torch.zeros(4, dtype=torch.int32, device=dev)
        flags = (ctypes.c_int32 * 4)()
        r = None
        if r_ij is not None and self.n_edges > 0:
            r = r_ij.detach().float().contiguous()
        with torch.cuda.device(dev):
            check(lib().spk_edge_plan(iptr(self.idx_i), iptr(self.idx_j), fptr(r), self.n_edges,
                                      self.n_atoms, iptr(self.rowptr, torch.int32),
                                      iptr(self.rev, torch.int32), iptr(scratch, torch.int32), flags, stream()))
        self.sorted = bool(flags[0])
        self.symmetric = bool(flags[2])
        n_half = 0
        if self.symmetric and self.n_edges > 0:
            # canonical edge of every undirected pair (e < rev[e]); one-off compaction per list
            ar = torch.arange(self.n_edges, dtype=torch.int32, device=dev)
            self.half = torch.nonzero(self.rev[: self.n_edges] > ar).flatten().to(torch.int32).contiguous()
            n_half = int(self.half.shape[0])
            if 2 * n_half != self.n_edges:
                self.symmetric = False
                self.half, n_half = None, 0
        self.groups = None
        self.edge_pair = None
        n_groups = max_ga = n_tiles_g = 0
        if self.symmetric and n_half > 0:
            # position in `half` of the pair of every directed edge (molecule-resident SchNet kernels)
            k = torch.arange(n_half, dtype=torch.int32, device=dev)
            self.edge_pair = torch.empty(self.n_edges, dtype=torch.int32, device=dev)
            self.edge_pair[self.half.long()] = k
            self.edge_pair[self.rev[: self.n_edges][self.half.long()].long()] = k
        if want_groups and self.symmetric and n_half > 0:   # block-diagonal structure: molecule-resident / group-local kernels
            self.groups = _block_diagonal_groups(self.idx_i, self.idx_j, self.half, self.n_atoms)
            if self.groups is not None:
                n_groups = int(self.groups[0].shape[0]) - 1
                max_ga = int(self.groups[3])
                n_tiles_g = int(self.groups[4])
        gp = self.groups
        self._graph = _lib.GraphT(self.n_atoms, self.n_edges, iptr(self.idx_i), iptr(self.idx_j),
                                  iptr(self.rowptr, torch.int32) if self.sorted else None,
                                  int(self.sorted), int(self.symmetric),
                                  iptr(self.rev, torch.int32) if self.symmetric else None,
                                  iptr(self.half, torch.int32) if self.half is not None else None, n_half,
                                  iptr(gp[0], torch.int32) if gp else None, iptr(gp[1], torch.int32) if gp else None,
                                  iptr(gp[2], torch.int32) if gp else None, n_groups, max_ga, n_tiles_g, 0, 0, None,
                                  iptr(self.edge_pair, torch.int32) if self.edge_pair is not None else None,
                                  int(gp[5]) if gp else 0, 0)

    def graph(self):
        return ctypes.byref(self._graph)

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


_MAX_GROUP_ATOMS = 32    # one 32-row MFMA tile of atoms per group (molecule-resident kernels, spk_schnet_mol.hip)


def _block_diagonal_groups(idx_i, idx_j, half, n_atoms):
    """Block-diagonal structure of a symmetric neighbour list (plan time, one small D2H sync):
    connected ranges of atoms that no edge leaves (molecules of a batch), merged greedily into groups of
    at most 32 atoms.  Returns (atom0 [G+1],
    pair0 [G+1], tile0 [G+1] int32 device tensors, max atoms per group, total tiles) or None when the
    list is not block diagonal with small blocks."""
    dev = idx_i.device
    ar = torch.arange(n_atoms, device=dev)
    mj = ar.clone()
    mj.scatter_reduce_(0, idx_i, idx_j, reduce="amax", include_self=True)
    cm = torch.cummax(mj, 0).values
    ends = torch.nonzero(cm == ar).flatten() + 1          # a component ends after every such atom
    ends_h = ends.cpu()
    sizes = torch.diff(ends_h, prepend=torch.zeros(1, dtype=ends_h.dtype))
    if sizes.numel() == 0 or int(sizes.max()) > _MAX_GROUP_ATOMS:
        return None
    cap = _MAX_GROUP_ATOMS
    atom0 = [0]
    cur = 0
    for sz in sizes.tolist():
        if cur + sz > cap and cur > 0:
            atom0.append(atom0[-1] + cur)
            cur = 0
        cur += sz
    atom0.append(atom0[-1] + cur)
    atom0_t = torch.tensor(atom0, dtype=torch.int64, device=dev)
    hi = idx_i[half.long()]                                # centre atom of every canonical pair (ascending)
    pair0 = torch.searchsorted(hi, atom0_t).to(torch.int32)
    tiles = (torch.diff(pair0.long()) + 31) // 32
    tile0 = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(tiles, 0)]).to(torch.int32)
    max_atoms = int(torch.diff(atom0_t).max())
    max_pairs = int(torch.diff(pair0.long()).max()) if pair0.numel() > 1 else 0
    return (atom0_t.to(torch.int32).contiguous(), pair0.contiguous(), tile0.contiguous(), max_atoms, int(tile0[-1]), max_pairs)


_PLAN_CACHE = collections.OrderedDict()
_PLAN_CACHE_SIZE = 16


def edge_plan(idx_i, idx_j, n_atoms, r_ij=None):
    """Cached plan of a neighbour list, keyed by the identity/version of the index tensors."""
    want_groups = True
    key = (idx_i.data_ptr(), idx_j.data_ptr(), idx_i._version, idx_j._version,
           int(idx_i.shape[0]), int(n_atoms), str(idx_i.device), r_ij is not None, want_groups)
    plan = _PLAN_CACHE.get(key)
    if plan is not None and plan._src[0] is idx_i and plan._src[1] is idx_j:
        _PLAN_CACHE.move_to_end(key)
        return plan
    plan = EdgePlan(idx_i, idx_j, n_atoms, r_ij, want_groups)
    plan._src = (idx_i, idx_j)  # keep the storage alive => data_ptr cannot be recycled
    _PLAN_CACHE[key] = plan
    while len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
        _PLAN_CACHE.popitem(last=False)
    return plan


class StaticLists:
    """Static-shape mode for HIP-graph replays of the differentiable (training) path.

    Plans are normally cached on the identity / version of the index tensors and validated with a host
    round trip -- neither survives a graph whose index BUFFERS are refilled between replays.  Inside
    ``with StaticLists() as sl:`` (and in the captured graph) every index tensor declared with
    ``sl.declare_sorted(idx, n_rows)`` gets its CSR row pointers from a device-only kernel launched by
    ``sl.refresh()`` (capture that call at the start of the step); all other indices take the atomic
    scatter; neighbour-list plans (symmetry, reverse map) are not used.  ``sl.check()`` polls the device
    flag that the refresh kernels raise when a declared index was not ascending / in range."""

    def __init__(self):
        self.entries = {}
        self.err = None

    def declare_sorted(self, idx, n_rows):
        _lib.require_device(idx)
        if idx.dtype != torch.int64 or not idx.is_contiguous():
            raise SpkHipError("StaticLists.declare_sorted: needs a contiguous int64 tensor")
        if self.err is None:
            self.err = torch.zeros(1, dtype=torch.int32, device=idx.device)
        self.entries[id(idx)] = (idx, int(n_rows), torch.zeros(int(n_rows) + 1, dtype=torch.int32, device=idx.device))

    def refresh(self):
        for idx, n_rows, rowptr in self.entries.values():
            with torch.cuda.device(idx.device):
                check(lib().spk_segment_rowptr_i32(iptr(idx), int(idx.shape[0]), n_rows, iptr(rowptr, torch.int32),
                                                   iptr(self.err, torch.int32), stream()))

    def rowptr(self, idx, dim_size):
        e = self.entries.get(id(idx))
        return e[2] if (e is not None and e[0] is idx and e[1] == int(dim_size)) else None

    def check(self):
        if self.err is not None:
            f = int(self.err.item())
            if f:
                self.err.zero_()
                raise SpkHipError("StaticLists: a declared index was %s" % ("not ascending" if f & 1 else "out of range"))

    def __enter__(self):
        global _STATIC
        self._prev = _STATIC
        _STATIC = self
        return self

    def __exit__(self, *exc):
        global _STATIC
        _STATIC = self._prev
        return False


_STATIC = None


def segment_rowptr(idx, dim_size):
    """rowptr tensor if ``idx`` is ascending, else None (cached; device-only in static-shape mode)."""
    if _STATIC is not None:
        return _STATIC.rowptr(idx, dim_size)
    plan = edge_plan(idx, idx, dim_size, None)
    return plan.rowptr if plan.sorted else None


# ----------------------------------------------------------------------------- scatter / gather
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


class ScatterAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, dim_size, dim, rowptr):
        ctx.save_for_backward(idx)
        ctx.dim = dim
        ctx.rowptr = rowptr
        return _scatter_raw(x, idx, dim_size, dim, rowptr)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        return GatherFn.apply(gy, idx, ctx.dim, ctx.rowptr), None, None, None, None


class GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, dim, rowptr):
        ctx.save_for_backward(idx)
        ctx.dim = dim
        ctx.rows = int(x.shape[dim])
        ctx.rowptr = rowptr
        return _gather_raw(x, idx, dim)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        rp = ctx.rowptr
        if rp is not None and rp.shape[0] != ctx.rows + 1:
            rp = None
        return ScatterAddFn.apply(gy, idx, ctx.rows, ctx.dim, rp), None, None, None


def _check_float(x, who):
    if not x.is_cuda:
        raise SpkHipError("%s: tensor on %s -- schnetpack_amd runs on ROCm devices only (no CPU "
                          "fallback)" % (who, x.device))
    if x.dtype != torch.float32:
        raise SpkHipError("%s: dtype %s unsupported; the HIP path computes in float32" % (who, x.dtype))


def scatter_add(x, idx_i, dim_size, dim=0):
    """nn/scatter.py:7-34 -- sum over values with the same index (HIP, differentiable to any
    order through ``gather``)."""
    _check_float(x, "scatter_add")
    idx = idx_i.long().contiguous()
    rowptr = segment_rowptr(idx, int(dim_size)) if idx.shape[0] > 0 else None
    return ScatterAddFn.apply(x, idx, int(dim_size), int(dim), rowptr)


def gather(x, idx, dim=0, rowptr=None):
    """x.index_select(dim, idx) on the HIP path (the transpose of scatter_add)."""
    _check_float(x, "gather")
    return GatherFn.apply(x, idx.long().contiguous(), int(dim), rowptr)


# ----------------------------------------------------------------------------- pairwise vectors
class PairwiseFn(torch.autograd.Function):
    """r_ij = R[idx_j] - R[idx_i] (+ offsets)  (atomistic/distances.py:14-26).  Backward scatters
    dL/dr_ij onto the atoms in one kernel; linear, so it is differentiable to any order through its
    transpose ``PairwiseBwdFn``."""

    @staticmethod
    def forward(ctx, R, idx_i, idx_j, offsets):
        Rc = R.contiguous()
        E = int(idx_i.shape[0])
        r = torch.empty((E, 3), dtype=torch.float32, device=R.device)
        oc = offsets.contiguous() if offsets is not None else None
        with torch.cuda.device(R.device):
            check(lib().spk_pairwise_f32(fptr(Rc), iptr(idx_i), iptr(idx_j), fptr(oc), E, fptr(r), stream()))
        ctx.save_for_backward(idx_i, idx_j)
        ctx.n = int(R.shape[0])
        ctx.has_off = offsets is not None
        # the plan of the list (cached; the representation asks for the same one): on symmetric sorted
        # lists the backward is a segmented row sum instead of 6 atomics per edge
        ctx.plan = edge_plan(idx_i, idx_j, ctx.n, r) if (E > 0 and ctx.needs_input_grad[0] and _STATIC is None) else None
        return r

    @staticmethod
    def backward(ctx, gr):
        idx_i, idx_j = ctx.saved_tensors
        gR = PairwiseBwdFn.apply(gr, idx_i, idx_j, ctx.n, ctx.plan) if ctx.needs_input_grad[0] else None
        goff = gr if (ctx.has_off and ctx.needs_input_grad[3]) else None
        return gR, None, None, goff


class PairwiseBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gr, idx_i, idx_j, n_atoms, plan=None):
        grc = gr.contiguous()
        gR = torch.empty((n_atoms, 3), dtype=torch.float32, device=gr.device)
        with torch.cuda.device(gr.device):
            if plan is not None:
                check(lib().spk_pairwise_bwd_graph_f32(fptr(grc), plan.graph(), fptr(gR), stream()))
            else:
                check(lib().spk_pairwise_bwd_f32(fptr(grc), iptr(idx_i), iptr(idx_j), int(idx_i.shape[0]), int(n_atoms), fptr(gR), stream()))
        ctx.save_for_backward(idx_i, idx_j)
        return gR

    @staticmethod
    def backward(ctx, ggR):
        idx_i, idx_j = ctx.saved_tensors
        return PairwiseFn.apply(ggR, idx_i, idx_j, None), None, None, None, None


def pairwise_vectors(R, idx_i, idx_j, offsets=None):
    _check_float(R, "pairwise_vectors")
    return PairwiseFn.apply(R, idx_i.long().contiguous(), idx_j.long().contiguous(), offsets)


# ----------------------------------------------------------------------------- radial / cutoff
def radial_struct(kind, n_rbf, p0, p1, cutoff):
    return _lib.RadialT(int(kind), int(n_rbf), fptr(p0), fptr(p1) if p1 is not None else None, float(cutoff))


class RadialCutoffFn(torch.autograd.Function):
    """(phi [.., n_rbf], fcut [..]) of distances; first-order backward on the HIP kernel."""

    @staticmethod
    def forward(ctx, d, kind, p0, p1, cutoff, want_phi, want_cut):
        dc = d.contiguous()
        n = dc.numel()
        n_rbf = int(p0.shape[0])
        rb = radial_struct(kind, n_rbf, p0, p1, cutoff)
        phi = torch.empty(tuple(d.shape) + (n_rbf,), dtype=torch.float32, device=d.device) if want_phi else None
        fc = torch.empty(d.shape, dtype=torch.float32, device=d.device) if want_cut else None
        with torch.cuda.device(d.device):
            check(lib().spk_radial_cutoff_f32(fptr(dc), n, ctypes.byref(rb), fptr(phi), fptr(fc), stream()))
        ctx.save_for_backward(dc, p0, p1 if p1 is not None else p0)
        ctx.meta = (kind, cutoff, p1 is not None)
        if want_phi and want_cut:
            return phi, fc
        return phi if want_phi else fc

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        dc, p0, p1 = ctx.saved_tensors
        kind, cutoff, has_p1 = ctx.meta
        rb = radial_struct(kind, int(p0.shape[0]), p0, p1 if has_p1 else None, cutoff)
        gphi = gfc = None
        for g in grads:
            if g is None:
                continue
            if g.dim() == dc.dim() + 1:
                gphi = g.contiguous()
            else:
                gfc = g.contiguous()
        gd = torch.empty_like(dc)
        with torch.cuda.device(dc.device):
            check(lib().spk_radial_cutoff_bwd_f32(fptr(dc), dc.numel(), ctypes.byref(rb), fptr(gphi), fptr(gfc), fptr(gd), stream()))
        return gd, None, None, None, None, None, None


# ----------------------------------------------------------------------------- dense
_ACT_IDS = {None: _lib.SPK_ACT_NONE, "none": _lib.SPK_ACT_NONE, "ssp": _lib.SPK_ACT_SSP, "silu": _lib.SPK_ACT_SILU}


def _act_grad(pre, act):
    if act == _lib.SPK_ACT_SSP:
        return torch.sigmoid(pre)
    if act == _lib.SPK_ACT_SILU:
        s = torch.sigmoid(pre)
        return s * (1.0 + pre * (1.0 - s))
    return None


def dense_raw(x, w, b, act, res=None, want_pre=False):
    """y = act(x w^T + b) (+ res) on the HIP kernels; x: [..., k]."""
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


class DenseFn(torch.autograd.Function):
    """nn/base.py:52-55.  Forward: HIP (fp32 MFMA when k % 8 == 0 and n_out % 32 == 0).
    Backward: differentiable torch algebra.  Under ``create_graph=True`` (training on forces) the
    pre-activation is re-derived from the graph tensors so that the second order (act'' terms)
    is exact."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        y, pre = dense_raw(x, w, b, act, want_pre=(act != _lib.SPK_ACT_NONE))
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, b if b is not None else w, pre if pre is not None else y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, b, pre = ctx.saved_tensors
        g = gy
        if ctx.act != _lib.SPK_ACT_NONE:
            if torch.is_grad_enabled():
                pre = torch.nn.functional.linear(x, w, b if ctx.has_bias else None)
            g = gy * _act_grad(pre, ctx.act)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g @ w
        if ctx.needs_input_grad[1]:
            gw = g.reshape(-1, g.shape[-1]).t() @ x.reshape(-1, x.shape[-1])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.reshape(-1, g.shape[-1]).sum(0)
        return gx, gw, gb, None


class DenseEvalFn(torch.autograd.Function):
    """Eval-mode Dense: forward and the first-order input gradient both on the HIP kernels; the
    weights are not differentiated (``Dense`` passes them detached when ``training`` is False)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        y, pre = dense_raw(x, w, b, act, want_pre=(act != _lib.SPK_ACT_NONE))
        ctx.act = act
        ctx.save_for_backward(w, pre if pre is not None else y)
        ctx.k = int(x.shape[-1])
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        w, pre = ctx.saved_tensors
        n_out = int(w.shape[0])
        g2 = gy.contiguous().view(-1, n_out)
        m = int(g2.shape[0])
        dx = torch.empty((m, ctx.k), dtype=torch.float32, device=gy.device)
        with torch.cuda.device(gy.device):
            check(lib().spk_dense_bwd_input_f32(fptr(g2), fptr(pre.view(-1, n_out)) if ctx.act != _lib.SPK_ACT_NONE else None,
                                                fptr(w.contiguous()), None, fptr(dx), m, ctx.k, n_out, int(ctx.act), stream()))
        return dx.view(tuple(gy.shape[:-1]) + (ctx.k,)), None, None, None


def dense(x, w, b=None, act=None, training=True):
    _check_float(x, "dense")
    a = _ACT_IDS[act] if not isinstance(act, int) else act
    if not training:
        return DenseEvalFn.apply(x, w.detach(), b.detach() if b is not None else None, a)
    return DenseFn.apply(x, w, b, a)


# ----------------------------------------------------------------------------- fused Atomwise head
class AtomwiseFn(torch.autograd.Function):
    """E_m = sum_{n in m} (w2 . act(W1 x_n + b1) + b2)  (atomistic/atomwise.py:69-88 with the default
    2-layer head), eval regime: one launch forward, one launch for the first-order gradient w.r.t. x;
    the weights are not differentiated.  Returns (E [n_mol], y_atom [N, 1])."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, idx_m, n_mol, act):
        xc = x.contiguous()
        N, n_in = int(xc.shape[0]), int(xc.shape[1])
        H = int(w1.shape[0])
        dev = x.device
        pre = torch.empty((N, H), dtype=torch.float32, device=dev)
        y_atom = torch.empty((N, 1), dtype=torch.float32, device=dev)
        E = torch.empty((int(n_mol),), dtype=torch.float32, device=dev)
        w1c, w2c = w1.contiguous(), w2.contiguous().view(-1)
        with torch.cuda.device(dev):
            check(lib().spk_atomwise_fwd_f32(fptr(xc), fptr(w1c), fptr(b1), fptr(w2c), fptr(b2), iptr(idx_m),
                                             N, n_in, H, int(act), int(n_mol), fptr(pre), fptr(y_atom), fptr(E), stream()))
        ctx.save_for_backward(pre, w1c, w2c, idx_m)
        ctx.meta = (N, n_in, H, int(act), int(n_mol))
        return E, y_atom

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gE, gy):
        pre, w1c, w2c, idx_m = ctx.saved_tensors
        N, n_in, H, act, n_mol = ctx.meta
        gx = torch.empty((N, n_in), dtype=torch.float32, device=pre.device)
        with torch.cuda.device(pre.device):
            check(lib().spk_atomwise_bwd_f32(fptr(gE.contiguous()), fptr(gy.contiguous().view(-1)), fptr(pre), fptr(w1c), fptr(w2c),
                                             iptr(idx_m), N, n_in, H, act, n_mol, fptr(gx), stream()))
        return (gx,) + (None,) * 7


def atomwise_supported(n_in, n_hidden, act):
    return bool(lib().spk_atomwise_supported(int(n_in), int(n_hidden), int(act)))


# ----------------------------------------------------------------------------- fused SchNet
class SchNetFn(torch.autograd.Function):
    """scalar_representation = SchNet(x0, r_ij) (representation/schnet.py:147-173), fused.
    Backward returns dL/dx0 and dL/dr_ij only (eval-mode force path)."""

    @staticmethod
    def forward(ctx, x0, r_ij, plan, rb_args, model_struct, keep):
        N, F = int(x0.shape[0]), int(x0.shape[1])
        dev = x0.device
        x0c = x0.contiguous()
        rc = r_ij.contiguous()
        L = lib()
        out = torch.empty((N, F), dtype=torch.float32, device=dev)
        rb = radial_struct(*rb_args)
        # keep the raw filter outputs for the backward when a gradient w.r.t. the geometry will be asked for
        model_struct.reserved = 1 if r_ij.requires_grad else 0
        if model_struct.reserved:
            n_saved = int(L.spk_schnet_saved_floats_graph(ctypes.byref(model_struct), plan.graph(), ctypes.byref(rb)))
        else:
            n_saved = int(L.spk_schnet_saved_floats(ctypes.byref(model_struct), N))
        saved = torch.empty(max(1, n_saved), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(1, int(L.spk_schnet_scratch_floats(ctypes.byref(model_struct), N))), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.spk_schnet_forward_f32(ctypes.byref(model_struct), plan.graph(), ctypes.byref(rb), fptr(x0c), fptr(rc),
                                           fptr(out), fptr(saved), fptr(scratch), stream()))
        ctx.save_for_backward(rc, saved)
        ctx.plan, ctx.rb_args, ctx.model_struct, ctx.keep = plan, rb_args, model_struct, keep
        ctx.scratch = scratch
        ctx.shape = (N, F)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx):
        rc, saved = ctx.saved_tensors
        N, F = ctx.shape
        dev = rc.device
        L = lib()
        gr = torch.empty_like(rc)
        gx0 = torch.empty((N, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        rb = radial_struct(*ctx.rb_args)
        with torch.cuda.device(dev):
            check(L.spk_schnet_backward_f32(ctypes.byref(ctx.model_struct), ctx.plan.graph(), ctypes.byref(rb),
                                            fptr(gx.contiguous()), fptr(rc), fptr(saved), fptr(ctx.scratch),
                                            fptr(gr), fptr(gx0), stream()))
        return gx0, (gr if ctx.needs_input_grad[1] else None), None, None, None, None


# ----------------------------------------------------------------------------- fused PaiNN
class PaiNNFn(torch.autograd.Function):
    """(scalar_representation, vector_representation) = PaiNN(q0, r_ij)
    (representation/painn.py:207-256), fused; first-order backward w.r.t. q0 and r_ij."""

    @staticmethod
    def forward(ctx, q0, r_ij, plan, rb_args, model_struct, keep):
        N, F = int(q0.shape[0]), int(q0.shape[1])
        dev = q0.device
        q0c = q0.contiguous()
        rc = r_ij.contiguous()
        L = lib()
        q = torch.empty((N, F), dtype=torch.float32, device=dev)
        mu = torch.empty((N, 3, F), dtype=torch.float32, device=dev)
        saved = torch.empty(max(1, int(L.spk_painn_saved_floats(ctypes.byref(model_struct), N))), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(1, int(L.spk_painn_scratch_floats(ctypes.byref(model_struct), N))), dtype=torch.float32, device=dev)
        rb = radial_struct(*rb_args)
        with torch.cuda.device(dev):
            check(L.spk_painn_forward_f32(ctypes.byref(model_struct), plan.graph(), ctypes.byref(rb), fptr(q0c), fptr(rc),
                                          fptr(q), fptr(mu), fptr(saved), fptr(scratch), stream()))
        ctx.save_for_backward(rc, saved)
        ctx.plan, ctx.rb_args, ctx.model_struct, ctx.keep = plan, rb_args, model_struct, keep
        ctx.scratch = scratch
        ctx.shape = (N, F)
        return q, mu

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gq, gmu):
        rc, saved = ctx.saved_tensors
        N, F = ctx.shape
        dev = rc.device
        L = lib()
        gr = torch.empty_like(rc)
        gq0 = torch.empty((N, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        rb = radial_struct(*ctx.rb_args)
        if gq is None and gmu is None:
            gq = torch.zeros((N, F), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.spk_painn_backward_f32(ctypes.byref(ctx.model_struct), ctx.plan.graph(), ctypes.byref(rb),
                                           fptr(gq.contiguous()) if gq is not None else None,
                                           fptr(gmu.contiguous()) if gmu is not None else None,
                                           fptr(rc), fptr(saved), fptr(ctx.scratch), fptr(gr), fptr(gq0), stream()))
        return gq0, (gr if ctx.needs_input_grad[1] else None), None, None, None, None
