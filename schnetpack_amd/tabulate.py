"""EXPERIMENT (round 3; eval only, never on by default): tabulated SchNet filters.

The filter of an interaction, ``W_l(d) f_c(d) = (ssp(phi(d) W1^T + b1) W2^T + b2) f_c(d)`` (representation/schnet.py:60-62), is a smooth
function of ONE variable per channel.  :func:`tabulate_filters` evaluates it and its slope in float64 on the host at ``n_knots``
equidistant distances in [0, cutoff] and attaches the tables to the interactions (``spk_filter_table_set``, keyed by the device
address of ``filter_network.1.weight``); the general SchNet driver then runs the convolution and its first-order backward through
the table kernels (``csrc/spk_tabfilter.hip``: cubic Hermite, one row pass, no atomics) instead of the fp32-MFMA filter network.
The fp32-MFMA path stays the contract path; measured time / error of both: ``scripts/tab_filter_experiment.py``,
``profiles/r03_tabulated_filter_experiment.json``, DESIGN.md section 7.

The tables are a snapshot of the weights: call :func:`tabulate_filters` again after the weights (or their device) changed,
:func:`clear_filter_tables` to go back.
"""
import math
import weakref
from typing import Optional

import torch

from . import _lib

# representation (weak) -> (tables tensor [L, n_knots, F, 4], device addresses the tables are registered under).  The addresses are kept as
# numbers: a model moved with .to() / re-allocated weights is detached by the addresses it WAS registered under, and a collected
# representation takes its entries with it (weakref finalizer) -- no stale address stays behind for a later allocation to alias.  The version
# of the weights travels with each table (spk_filter_table_set_stamp): the operator library drops a table whose weights changed in place.
_KEEP = weakref.WeakKeyDictionary()


def _detach(addresses):
    import ctypes
    for a in addresses:
        _lib.lib().spk_filter_table_set(ctypes.c_void_p(a), None, 0, 0.0)


def _remember(rep, table, addresses):
    """One live finalizer per representation: the handle is kept with the entry and detached (cancelled) when the entry is replaced or
    cleared -- a finalizer of an EARLIER registration would otherwise fire at collection time with addresses the allocator may since have
    handed to another model's weights, and silently drop that model's tables."""
    old = _KEEP.pop(rep, None)
    if old is not None and len(old) > 2:
        old[2].detach()
    fin = weakref.finalize(rep, _detach, list(addresses))
    _KEEP[rep] = (table, list(addresses), fin)


def _refuse_trainable(rep):
    if getattr(rep.radial_basis, "trainable", False):
        raise _lib.SpkHipError("tabulate_filters: the radial basis is trainable -- the tables are a snapshot keyed on the versions of the filter-network weights "
                               "only, offsets / widths that change under them would leave a stale table; tabulate a model with a fixed basis")




def pack_knots(v: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """Table rows from float64 knot values ``v`` and slopes x step ``m`` ([n_knots, C] each): per knot n and channel the four
    floats (v_n, m_n, v_{n+1} - v_n, m_{n+1}) -- everything an interval needs in ONE 16-byte read.  The DIFFERENCE of neighbouring
    knot values is formed in float64 before rounding: taken from two rounded float32 values it carries their rounding error divided
    by the knot spacing into the slope (4e-6 of max |dW/dd| at 512 knots, which bounded the forces at 3e-6 of the contract path
    before this layout)."""
    dv = torch.zeros_like(v)
    dv[:-1] = v[1:] - v[:-1]
    m1 = torch.zeros_like(m)
    m1[:-1] = m[1:]
    return torch.stack([v, m, dv, m1], -1)


def filter_and_slope(interaction, radial_basis, cutoff: float, d: torch.Tensor):
    """(W f_c, d(W f_c)/dd) at distances ``d`` ([n]) in float64 on the host, analytic derivative."""
    d = d.double().cpu()
    kind, p0, p1 = radial_basis.kernel_params()
    p0 = p0.double().cpu()
    if int(kind) == _lib.SPK_RBF_GAUSSIAN:
        c = -0.5 / p1.double().cpu() ** 2
        t = d[:, None] - p0[None, :]
        phi = torch.exp(c * t * t)
        dphi = 2.0 * c * t * phi
    else:
        arg = d[:, None] * p0[None, :]
        inv = torch.where(d == 0, torch.ones_like(d), 1.0 / d)[:, None]
        phi = torch.sin(arg) * inv
        dphi = (p0[None, :] * torch.cos(arg) - phi) * inv
    w1, b1 = interaction.filter_network[0].weight.detach().double().cpu(), interaction.filter_network[0].bias.detach().double().cpu()
    w2, b2 = interaction.filter_network[1].weight.detach().double().cpu(), interaction.filter_network[1].bias.detach().double().cpu()
    a = phi @ w1.t() + b1
    hid = torch.nn.functional.softplus(a) - math.log(2.0)
    dhid = torch.sigmoid(a) * (dphi @ w1.t())
    g = hid @ w2.t() + b2
    dg = dhid @ w2.t()
    inside = (d < cutoff).double()
    fc = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0) * inside
    dfc = -0.5 * math.pi / cutoff * torch.sin(d * math.pi / cutoff) * inside
    return g * fc[:, None], dg * fc[:, None] + g * dfc[:, None]


def _basis(radial_basis, d: torch.Tensor):
    kind, p0, p1 = radial_basis.kernel_params()
    p0 = p0.double().cpu()
    if int(kind) == _lib.SPK_RBF_GAUSSIAN:
        c = -0.5 / p1.double().cpu() ** 2
        t = d[:, None] - p0[None, :]
        phi = torch.exp(c * t * t)
        return phi, 2.0 * c * t * phi
    arg = d[:, None] * p0[None, :]
    inv = torch.where(d == 0, torch.ones_like(d), 1.0 / d)[:, None]
    phi = torch.sin(arg) * inv
    return phi, (p0[None, :] * torch.cos(arg) - phi) * inv


def _tabulate_painn(rep, n_knots: int) -> torch.Tensor:
    """PaiNN (painn.py:232-236): the RAW filter of an interaction, phi(d) W_f^T + b_f over its 3F rows (the cutoff stays in the
    kernels), attached to the address of those rows of ``filter_net.weight``."""
    F = rep.n_atom_basis
    if F != 128 or not getattr(rep, "_fused", False):
        raise _lib.SpkHipError("tabulate_filters: the table kernels cover n_atom_basis = 128 and the fused PaiNN path only")
    cutoff = float(rep.cutoff_fn.cutoff_value())
    d = torch.linspace(0.0, cutoff, int(n_knots), dtype=torch.float64)
    step = cutoff / (int(n_knots) - 1)
    phi, dphi = _basis(rep.radial_basis, d)
    w, b = rep.filter_net.weight.detach().double().cpu(), rep.filter_net.bias.detach().double().cpu()
    L = len(rep.interactions)
    shared = w.shape[0] == 3 * F
    tabs = []
    for l in range(L):
        rows = slice(0, 3 * F) if shared else slice(3 * F * l, 3 * F * (l + 1))
        tabs.append(pack_knots(phi @ w[rows].t() + b[rows], (dphi @ w[rows].t()) * step))
    dev = rep.filter_net.weight.device
    table = torch.stack(tabs).float().contiguous().to(dev)
    keys = []
    for l in range(L):
        rows = slice(0, 3 * F) if shared else slice(3 * F * l, 3 * F * (l + 1))
        key = rep.filter_net.weight.detach()[rows]
        _lib.check(_lib.lib().spk_filter_table_set(_lib.fptr(key), _lib.fptr(table[l]), int(n_knots), cutoff))
        _lib.check(_lib.lib().spk_filter_table_set_stamp(_lib.fptr(key), 1 + rep.filter_net.weight._version + rep.filter_net.bias._version))
        keys.append(key.data_ptr())
    _remember(rep, table, keys)
    torch.ops.spk_hip.clear_caches()
    return table


def tabulate_filters(representation, n_knots: Optional[int] = None) -> torch.Tensor:
    """Build and attach the filter tables of every interaction of a SchNet or PaiNN representation (on its device).  Returns the
    tables ``[n_interactions, n_knots, n_filters (3 n_atom_basis for PaiNN), 4]`` (:func:`pack_knots`).  ``n_knots=None``: 512 for Gaussian
    bases, 1024 for Bessel bases."""
    rep = representation
    if n_knots is None:          # Bessel filters oscillate faster: 512 knots leave 1.5e-5 of max |dW/dd| in the slope, 1024 knots 1.6e-6
        kind = int(rep.radial_basis.kernel_params()[0])
        n_knots = 512 if kind == _lib.SPK_RBF_GAUSSIAN else 1024
    _refuse_trainable(rep)
    clear_filter_tables(rep)
    if hasattr(rep, "filter_net"):
        return _tabulate_painn(rep, n_knots)
    if rep.n_filters != 128 or not rep._fused:
        raise _lib.SpkHipError("tabulate_filters: the table kernels cover n_filters = 128 and the fused filter network (ssp) only")
    cutoff = float(rep.cutoff_fn.cutoff_value())
    d = torch.linspace(0.0, cutoff, int(n_knots), dtype=torch.float64)
    step = cutoff / (int(n_knots) - 1)
    tabs = []
    for it in rep.interactions:
        W, dW = filter_and_slope(it, rep.radial_basis, cutoff, d)
        tabs.append(pack_knots(W, dW * step))
    dev = rep.interactions[0].filter_network[1].weight.device
    table = torch.stack(tabs).float().contiguous().to(dev)
    keys = []
    for l, it in enumerate(rep.interactions):
        w2 = it.filter_network[1].weight
        _lib.check(_lib.lib().spk_filter_table_set(_lib.fptr(w2.detach()), _lib.fptr(table[l]), int(n_knots), cutoff))
        fn = it.filter_network
        _lib.check(_lib.lib().spk_filter_table_set_stamp(_lib.fptr(w2.detach()), 1 + fn[0].weight._version + fn[0].bias._version + w2._version + fn[1].bias._version))
        keys.append(w2.data_ptr())
    _remember(rep, table, keys)
    torch.ops.spk_hip.clear_caches()          # parameter blocks are rebuilt: the drivers look the tables up by weight address
    return table


def clear_filter_tables(representation: Optional[object] = None):
    """Detach the tables of one representation (or of all): the fp32-MFMA filter network runs again."""
    if representation is None:
        _lib.lib().spk_filter_table_clear()
        for ent in list(_KEEP.values()):
            if len(ent) > 2:
                ent[2].detach()
        _KEEP.clear()
        return
    ent = _KEEP.pop(representation, None)
    if ent is not None:
        if len(ent) > 2:
            ent[2].detach()
        _detach(ent[1])
