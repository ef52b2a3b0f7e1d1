"""The force-matching gradient ENGINE (schnetpack_amd/csrc/spk_fm_engine.h + spk_fm_kernels.h: the host orchestration and the kernel
bodies that the HIP build runs) instantiated on a serial CPU backend (tests/fm_emu, test infrastructure) and compared, in float64,
with the pinned restatement oracle/fm_oracle.py: energies, forces and every weight gradient of the force-matching loss
(atomistic/response.py:59-68 with create_graph = training; task.py:166-185).  A wrong buffer, stride, stacking offset or formula in
the engine shows up here without a GPU; the device build is then compared with the same oracle in tests/test_gpu_fm.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import fm_oracle as FM
from oracle import spk_oracle as O
from schnetpack_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "fm_emu")
CSRC = os.path.join(HERE, "..", "schnetpack_amd", "csrc")


class EmuDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("kind", "F", "nf", "L", "K", "H", "head_act", "rbf_kind", "shared", "n_types")] + \
               [("cutoff", ctypes.c_double), ("eps", ctypes.c_double), ("N", ctypes.c_int64), ("E", ctypes.c_int64), ("M", ctypes.c_int64)]


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(EMU_DIR, "fm_emu.cpp")
    lib = os.path.join(EMU_DIR, "libspk_fm_emu.so")
    deps = [src, os.path.join(CSRC, "spk_fm_engine.h"), os.path.join(CSRC, "spk_fm_kernels.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", src, "-o", lib])
    L = ctypes.CDLL(lib)
    L.fm_emu_ws_bytes.restype = ctypes.c_int64
    L.fm_emu_grad_floats.restype = ctypes.c_int64
    return L


SCHNET_KEYS = ["in2f.weight", "filter_network.0.weight", "filter_network.0.bias", "filter_network.1.weight", "filter_network.1.bias",
               "f2out.0.weight", "f2out.0.bias", "f2out.1.weight", "f2out.1.bias"]
PAINN_KEYS = [("interactions.%d.interatomic_context_net.0.weight"), ("interactions.%d.interatomic_context_net.0.bias"),
              ("interactions.%d.interatomic_context_net.1.weight"), ("interactions.%d.interatomic_context_net.1.bias"),
              ("mixing.%d.mu_channel_mix.weight"), ("mixing.%d.intraatomic_context_net.0.weight"), ("mixing.%d.intraatomic_context_net.0.bias"),
              ("mixing.%d.intraatomic_context_net.1.weight"), ("mixing.%d.intraatomic_context_net.1.bias")]
HEAD_KEYS = ["outnet.0.weight", "outnet.0.bias", "outnet.1.weight", "outnet.1.bias"]


def weight_names(kind, L):
    """Names in the order of the engine's flat gradient layout (spk_fm_engine.h)."""
    if kind == "schnet":
        names = ["interactions.%d.%s" % (l, k) for l in range(L) for k in SCHNET_KEYS]
    else:
        names = [k % l for l in range(L) for k in PAINN_KEYS] + ["filter_net.weight", "filter_net.bias"]
    return names + HEAD_KEYS + ["embedding.weight"]


def run_emu(emu, kind, rep_p, head_p, b, L, gfun, f64=True, shared=False, eps=1e-8):
    dt = np.float64 if f64 else np.float32
    allp = dict(rep_p)
    allp.update(head_p)
    names = weight_names(kind, L)
    arrs = [np.ascontiguousarray(allp[k].numpy().astype(dt)) for k in names]
    if "radial_basis.freqs" in rep_p:
        p0, p1, rk = rep_p["radial_basis.freqs"], rep_p["radial_basis.freqs"], 1
    else:
        p0, p1, rk = rep_p["radial_basis.offsets"], rep_p["radial_basis.widths"], 0
    arrs += [np.ascontiguousarray(p0.numpy().astype(dt)), np.ascontiguousarray(p1.numpy().astype(dt))]
    F_ = allp["embedding.weight"].shape[1]
    K = int(p0.shape[0])
    nf = allp["interactions.0.in2f.weight"].shape[0] if kind == "schnet" else F_
    N, E, M = int(b["Z"].shape[0]), int(b["idx_i"].shape[0]), int(b["n_mol"])
    d = EmuDesc(0 if kind == "schnet" else 1, F_, nf, L, K, head_p["outnet.0.weight"].shape[0], 2, rk, int(shared), allp["embedding.weight"].shape[0],
                float(rep_p["cutoff_fn.cutoff"]), eps, N, E, M)
    wv = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    Z, ii, jj, im = (np.ascontiguousarray(b[k].numpy().astype(np.int64)) for k in ("Z", "idx_i", "idx_j", "idx_m"))
    R, off = np.ascontiguousarray(b["R"].numpy().astype(dt)), np.ascontiguousarray(b["offsets"].numpy().astype(dt))
    ws = np.zeros(int(emu.fm_emu_ws_bytes(ctypes.byref(d), int(f64))) + 64, np.uint8)
    Eo, Fo = np.zeros(M, dt), np.zeros((N, 3), dt)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    rc = emu.fm_emu_forward(ctypes.byref(d), int(f64), wv, P(Z), P(ii), P(jj), P(im), P(R), P(off), P(ws), P(Eo), P(Fo))
    assert rc == 0, rc
    gE, gF = gfun(torch.from_numpy(Eo).double(), torch.from_numpy(Fo).double())
    gE, gF = np.ascontiguousarray(gE.numpy().astype(dt)), np.ascontiguousarray(gF.numpy().astype(dt))
    ng = int(emu.fm_emu_grad_floats(ctypes.byref(d)))
    grads = np.full(ng, np.nan, dt)
    rc = emu.fm_emu_backward(ctypes.byref(d), int(f64), wv, P(Z), P(ii), P(jj), P(im), P(R), P(off), P(ws), P(gE), P(gF), P(grads))
    assert rc == 0, rc
    out, o = {}, 0
    for k, a in zip(names, arrs):
        out[k] = torch.from_numpy(grads[o:o + a.size].reshape(a.shape).astype(np.float64))
        o += a.size
    assert o == ng
    return torch.from_numpy(Eo).double(), torch.from_numpy(Fo).double(), out


def _params(kind, F_, L, n_rbf, radial, shared, nf=None):
    if kind == "schnet":
        rep_p = O.init_schnet_params(F_, L, n_rbf, 5.0, radial=radial, seed=0, n_filters=nf)
    else:
        rep_p = O.init_painn_params(F_, L, n_rbf, 5.0, radial=radial, seed=0, shared_filters=shared)
    head_p = O.init_atomwise_params(F_, seed=1)
    torch.manual_seed(7)
    for p in (rep_p, head_p):
        for k in list(p):
            if k.endswith("bias"):
                p[k] = 0.1 * torch.randn_like(p[k])
    return rep_p, head_p


CASES = [("schnet", "gaussian", False, None), ("schnet", "bessel", False, 24), ("painn", "gaussian", False, None), ("painn", "bessel", False, None),
         ("painn", "gaussian", True, None)]


@pytest.mark.parametrize("kind,radial,shared,nf", CASES)
def test_engine_matches_the_restatement_in_float64(emu, kind, radial, shared, nf):
    F_, L, n_rbf = 16, 3, 8
    b = synthetic.molecule_batch("aspirin", n_frames=2, cutoff=5.0, seed=5)
    # a batch whose list is NOT symmetric and whose neighbour index is in no particular order (vesin / LAMMPS lists, pair_schnetpack.cpp:240-267)
    keep = torch.ones(b["idx_i"].shape[0], dtype=torch.bool)
    keep[::7] = False
    b = dict(b)
    for k in ("idx_i", "idx_j", "offsets"):
        b[k] = b[k][keep]
    rep_p, head_p = _params(kind, F_, L, n_rbf, radial, shared, nf)
    M, N = int(b["n_mol"]), b["Z"].shape[0]
    torch.manual_seed(11)
    Et, Ft = torch.randn(M, dtype=torch.float64), torch.randn(N, 3, dtype=torch.float64)
    gfun = lambda E, F: (2 * 0.01 * (E - Et) / M, 2 * 0.99 * (F - Ft) / (3 * N))
    if kind == "schnet":
        E_o, F_o, saved = FM.schnet_forward(rep_p, head_p, b, L)
        g_o = FM.schnet_backward(saved, *gfun(E_o, F_o))
    else:
        E_o, F_o, saved = FM.painn_forward(rep_p, head_p, b, L, shared_filters=shared)
        g_o = FM.painn_backward(saved, *gfun(E_o, F_o))
    # the engine sees the list padded with inert pairs (train.pad_edges: static-shape batches): the tail behind the last pair inside the
    # cutoff is skipped by the row / column loops (e_act) and must change nothing
    from schnetpack_amd.train import pad_edges
    bp = dict(b)
    bp["idx_i"], bp["idx_j"], bp["offsets"] = pad_edges(b["idx_i"], b["idx_j"], b["offsets"], N, b["idx_i"].shape[0] + 37, 5.0)
    E, F, g = run_emu(emu, kind, rep_p, head_p, bp, L, gfun, f64=True, shared=shared)
    assert torch.allclose(E, E_o, rtol=1e-12, atol=1e-12)
    assert torch.allclose(F, F_o, rtol=1e-10, atol=1e-12)
    for k, ref in g_o.items():
        err = float((g[k].reshape(ref.shape) - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)
        assert err < 1e-10, (k, err)
    assert set(g) == set(g_o)


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_engine_in_float32_at_the_model_width(emu, kind):
    """F = 128, 20 Gaussians, 3 interactions (the configs[3] model) in fp32 arithmetic against the float64 restatement:
    what the device build has to reproduce (weight gradients relative to each tensor's largest entry)."""
    F_, L, n_rbf = 128, 3, 20
    b = synthetic.molecule_batch("aspirin", n_frames=1, cutoff=5.0, seed=2)
    rep_p, head_p = _params(kind, F_, L, n_rbf, "gaussian", False)
    M, N = int(b["n_mol"]), b["Z"].shape[0]
    torch.manual_seed(12)
    Et, Ft = torch.randn(M, dtype=torch.float64), torch.randn(N, 3, dtype=torch.float64)
    gfun = lambda E, F: (2 * 0.01 * (E - Et) / M, 2 * 0.99 * (F - Ft) / (3 * N))
    if kind == "schnet":
        E_o, F_o, saved = FM.schnet_forward(rep_p, head_p, b, L)
        g_o = FM.schnet_backward(saved, *gfun(E_o, F_o))
    else:
        E_o, F_o, saved = FM.painn_forward(rep_p, head_p, b, L)
        g_o = FM.painn_backward(saved, *gfun(E_o, F_o))
    E, F, g = run_emu(emu, kind, rep_p, head_p, b, L, gfun, f64=False)
    assert float((E - E_o).abs().max()) / float(E_o.abs().max()) < 1e-5
    assert float((F - F_o).abs().max()) / float(F_o.abs().max()) < 1e-5
    worst = max(float((g[k].reshape(ref.shape) - ref).abs().max()) / (float(ref.abs().max()) + 1e-300) for k, ref in g_o.items())
    assert worst < 2e-5, worst
