"""TEST INFRASTRUCTURE ONLY: plain-torch float64 kernels for the raw ``spk_hip`` operators on the CPU dispatch key.

The product registers a loud refusal on that key (there is no CPU path).  The autograd layer of the operator library
(`spk_torch.cpp` / `spk_torch_train.h`: which operator each backward calls, with which roles, to which order) is host C++
that does not depend on the device, so the build box can check it without a GPU: inside ``with reference_kernels():`` the
refusals are overridden by the formulas below and ``torch.autograd.gradcheck`` / ``gradgradcheck`` run in float64 through the
real C++ autograd Functions.  Leaving the context restores the refusal.  The ``-m gpu`` tests compare the HIP kernels with
the same formulas evaluated by torch on the device.
"""
import contextlib
import math

import torch

from schnetpack_amd import torchops  # noqa: F401

LN2 = math.log(2.0)


def act_order(z, act, order):
    if act == 0:
        return z if order == 0 else (torch.ones_like(z) if order == 1 else torch.zeros_like(z))
    s = torch.sigmoid(z)
    s1 = s * (1 - s)
    if act == 1:
        return [torch.nn.functional.softplus(z) - LN2, s, s1, s1 * (1 - 2 * s), s1 * (1 - 6 * s + 6 * s * s)][order]
    return [z * s, s * (1 + z * (1 - s)), s1 * (2 + z * (1 - 2 * s)), s1 * (3 * (1 - 2 * s) + z * (1 - 6 * s + 6 * s * s))][order]


def act_mul(a, z, act, order, c=None):
    v = act_order(z, act, order)
    if a is not None:
        v = v * a
    if c is not None:
        v = v + c
    return v


def radial_order(d, kind, p0, p1, cutoff, order):
    """[..., R] derivative of the given order of the radial functions at d (kind 2: [...], the cosine cutoff)."""
    if kind == 0:
        c = -0.5 / p1 ** 2
        t = d[..., None] - p0
        phi = torch.exp(c * t * t)
        u = 2 * c * t
        return [phi, u * phi, (2 * c + u * u) * phi, (6 * c * u + u ** 3) * phi][order]
    if kind == 1:
        f = p0
        dd = d[..., None]
        q = 1.0 / torch.where(dd == 0, torch.ones_like(dd), dd)
        s, co = torch.sin(f * dd), torch.cos(f * dd)
        v = [s * q, (f * co - s * q) * q, ((2 * q * q - f * f) * s - 2 * f * q * co) * q,
             ((6 * q * q - f * f) * f * co + (3 * f * f - 6 * q * q) * q * s) * q][order]
        return torch.where(dd == 0, torch.zeros_like(v), v)
    a = math.pi / cutoff
    s, co = torch.sin(a * d), torch.cos(a * d)
    v = [0.5 * (co + 1), -0.5 * a * s, -0.5 * a * a * co, 0.5 * a ** 3 * s][order]
    return v * (d < cutoff).to(d.dtype)


def radial_d(d, a, kind, p0, p1, cutoff, order):
    v = radial_order(d, kind, p0, p1, cutoff, order)
    if a is not None:
        v = v * (a if kind == 2 else a[..., None])
    return v


def radial_c(G, d, a, kind, p0, p1, cutoff, order):
    v = (G * radial_order(d, kind, p0, p1, cutoff, order)).sum(-1)
    return v if a is None else v * a


def cfconv(x, W, idx_out, idx_src, n_out):
    xs = x if idx_src is None else x[idx_src]
    if idx_out is None:
        return xs * W
    return torch.zeros(n_out, W.shape[1], dtype=x.dtype, device=x.device).index_add(0, idx_out, xs * W)


def edge_mul(a, b, ia, ib):
    return (a if ia is None else a[ia]) * (b if ib is None else b[ib])


def vec3(op, A, B):
    """the five 3-vector products (include/spk_hip.h: SPK_VEC3_*); V: [..., 3, F], s: [..., F] (or [..., 1, F]), u: [..., 3]"""
    F = A.shape[-1]
    if op == 0:
        return A * B.reshape(A.shape[:-2] + (1, F))
    if op == 1:
        return (A * B).sum(-2, keepdim=True)
    if op == 2:
        return A.reshape(B.shape[:-1] + (1, F)) * B[..., None]
    if op == 3:
        return (A * B[..., None]).sum(-2, keepdim=True)
    return (A * B.reshape(A.shape[:-2] + (1, F))).sum(-1)


def dense_forward(x, w, b, act):
    pre = torch.nn.functional.linear(x, w, b)
    return (act_order(pre, act, 0), pre) if act != 0 else (pre, pre.new_empty(0))


def dense_backward_input(gy, pre, w, act):
    return (gy * act_order(pre, act, 1) if act != 0 else gy) @ w


def scatter_add(x, idx, dim_size, dim=0):
    shape = list(x.shape)
    shape[dim] = dim_size
    return torch.zeros(shape, dtype=x.dtype, device=x.device).index_add(dim, idx, x)


def pairwise(R, ii, jj, off):
    r = R[jj] - R[ii]
    return r if off is None else r + off


def pairwise_backward(gr, ii, jj, n):
    return torch.zeros(n, 3, dtype=gr.dtype, device=gr.device).index_add(0, jj, gr).index_add(0, ii, -gr)


KERNELS = {
    "act_mul": act_mul,
    "linear": lambda x, w, b: torch.nn.functional.linear(x, w, b),
    "matmul_nn": lambda u, w: u @ w,
    "matmul_tn": lambda u, x: (u.reshape(-1, u.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), u.reshape(-1, u.shape[-1]).sum(0)),
    "gemm_pair": lambda a, w, trans, u, x: ((a @ w) if trans else torch.nn.functional.linear(a, w),
                                            u.reshape(-1, u.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), u.reshape(-1, u.shape[-1]).sum(0)),
    "cfconv": cfconv,
    "edge_mul": edge_mul,
    "vec3": vec3,
    "radial_d": radial_d,
    "radial_c": radial_c,
    "rowscale": lambda W, s: W * s.reshape(W.shape[:-1])[..., None],
    "rowdot": lambda a, b: (a * b).sum(-1),
    "edge_norm": lambda r: torch.linalg.norm(r, dim=1),
    "dense_forward": dense_forward,
    "dense_backward_input": dense_backward_input,
    "scatter_add": scatter_add,
    "gather": lambda x, idx, dim=0: x.index_select(dim, idx),
    "pairwise": pairwise,
    "pairwise_backward": pairwise_backward,
}


_MIRRORS = ("schnetpack_amd.nn.scatter", "schnetpack_amd.nn.base", "schnetpack_amd.nn.cutoff", "schnetpack_amd.nn.radial",
            "schnetpack_amd.representation.schnet", "schnetpack_amd.representation.painn", "schnetpack_amd.atomistic")


@contextlib.contextmanager
def reference_kernels():
    """Test-only CPU kernels on the operators' CPU key.  The module mirrors would send host tensors down their plain-ATen route
    (schnetpack_amd/nn/fallback.py) and never reach the operators: inside this context their ``use_aten`` says "no", so that the host
    tensors travel through the operator library -- the thing under test."""
    import importlib
    lib = torch.library.Library("spk_hip", "IMPL")
    mods = [importlib.import_module(m) for m in _MIRRORS]
    saved = [getattr(m, "use_aten") for m in mods]
    try:
        for name, fn in KERNELS.items():
            lib.impl(name, fn, "CPU", allow_override=True)
        for m in mods:
            m.use_aten = lambda x: False
        yield
    finally:
        for m, f in zip(mods, saved):
            m.use_aten = f
        lib._destroy()
