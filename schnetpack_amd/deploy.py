"""Deployment of a trained potential without torch at run time.

The reference deploys by scripting the whole ``NeuralNetworkPotential`` (src/scripts/spkdeploy:16-40:
postprocessors minus the dtype casts, ``torch.jit.save`` with a ``cutoff`` metadata entry) and loads the
archive from LAMMPS with libtorch (interfaces/lammps/pair_schnetpack.cpp:128, :328).  The HIP kernels are
reached through a C ABI, not through TorchScript, so the deployed artefact here is a flat weight file that
``libspk_hip.so`` itself loads and runs (``spk_potential_*`` in include/spk_hip.h): a LAMMPS pair style, an
ASE calculator or any MD code links the one shared library and needs neither Python nor libtorch.

``export_potential(model, path)``  -- what ``spkdeploy model deployed`` does.
``DeployedPotential(path)``        -- ctypes handle on the runtime, numpy in / numpy out (used by the tests and
                                     as the template for the C++ side).

File layout (little endian): 8-byte magic ``SPKHIP01``; 16 int32 (version, kind, n_atom_basis, n_filters,
n_interactions, n_rbf, radial kind, head hidden width, head activation, embedding rows, is_extensive,
atomref rows, n_tensors, 3 reserved); 4 float32 (cutoff, epsilon, energy mean, reserved); n_tensors entries of
(32-byte zero padded name, int64 n_floats, int64 offset in floats from the data start); padding to a multiple of
64 bytes; the fp32 data.  Tensor names are the field names of ``spk_schnet_layer_t`` / ``spk_painn_layer_t``
prefixed with ``l<k>.``, plus ``embedding``, ``rbf_p0``, ``rbf_p1``, ``filt_w``, ``filt_b`` (PaiNN),
``head_w1``, ``head_b1``, ``head_w2``, ``head_b2`` and ``atomref``.
"""
import ctypes
import struct
from typing import Dict, Optional

import numpy as np

MAGIC = b"SPKHIP01"
VERSION = 1
KIND_SCHNET, KIND_PAINN = 0, 1

__all__ = ["export_potential", "DeployedPotential", "MAGIC"]


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().float().numpy())


def _find_head(model):
    heads = [m for m in model.output_modules if hasattr(m, "outnet")]
    forces = [m for m in model.output_modules if type(m).__name__ == "Forces"]
    if len(heads) != 1 or len(forces) != 1:
        raise ValueError("deploy: the model must have exactly one Atomwise head and one Forces module "
                         "(spkdeploy has the same restriction: forces by differentiation of the energy)")
    head, frc = heads[0], forces[0]
    if getattr(head, "aggregation_mode", "sum") != "sum" or getattr(head, "n_out", 1) != 1:
        raise ValueError("deploy: only a summed scalar Atomwise head is supported")
    if getattr(frc, "energy_key", "energy") != getattr(head, "output_key", "energy"):
        raise ValueError("deploy: Forces must differentiate the Atomwise output")
    if getattr(frc, "calc_stress", False):
        raise ValueError("deploy: stress is not part of the deployed force call")
    return head


def _offsets_of(model, energy_key):
    """(mean, is_extensive, atomref or None) from an ``AddOffsets`` postprocessor of the energy
    (transform/atomistic.py:217-324); dtype casts are dropped exactly like spkdeploy:21-23 does."""
    mean, ext, atomref = 0.0, 1, None
    for pp in list(getattr(model, "postprocessors", None) or []):
        name = type(pp).__name__
        if name in ("CastTo64", "CastTo32"):
            continue
        if name != "AddOffsets":
            raise ValueError("deploy: postprocessor %s is not supported" % name)
        if getattr(pp, "_property", energy_key) != energy_key:
            raise ValueError("deploy: AddOffsets on %r (only the energy can carry offsets)" % pp._property)
        ext = 1 if pp.is_extensive else 0
        if pp.add_mean:
            mean += float(pp.mean.detach().float().reshape(-1)[0])
        if pp.add_atomrefs:
            a = _np(pp.atomref).reshape(-1)
            atomref = a if atomref is None else atomref + a
    return mean, ext, atomref


def export_potential(model, path: Optional[str] = None) -> bytes:
    """Serialise ``model`` (a ``NeuralNetworkPotential`` with a SchNet / PaiNN representation from this package
    -- or a reference pickle loaded after ``schnetpack_amd.install`` -- an ``Atomwise`` energy head and ``Forces``)
    for ``spk_potential_load``.  Returns the bytes; writes them to ``path`` if given."""
    from . import _lib
    from .nn.base import activation_id
    rep = model.representation
    kind_name = type(rep).__name__
    if kind_name not in ("SchNet", "PaiNN"):
        raise ValueError("deploy: unsupported representation %s" % kind_name)
    if len(getattr(rep, "electronic_embeddings", [])) > 0:
        raise ValueError("deploy: electronic embeddings are not supported")
    if not rep._fusable():
        raise ValueError("deploy: this %s configuration has no fused HIP path (activation / trainable basis)" % kind_name)
    head = _find_head(model)
    net = head.outnet
    if len(net) != 2 or net[1].out_features != 1:
        raise ValueError("deploy: the head must be build_mlp(n_in, 1, n_layers=2) (atomwise.py:58-66)")
    head_act = activation_id(net[0].activation)
    if head_act not in (_lib.SPK_ACT_SSP, _lib.SPK_ACT_SILU) or activation_id(net[1].activation) != _lib.SPK_ACT_NONE:
        raise ValueError("deploy: unsupported head activation")
    F = int(rep.n_atom_basis)
    rbk, p0, p1 = rep.radial_basis.kernel_params()
    n_rbf, cutoff = int(rep.radial_basis.n_rbf), float(rep.cutoff_fn.cutoff_value())
    tensors: Dict[str, np.ndarray] = {}
    tensors["embedding"] = _np(rep.embedding.weight)
    tensors["rbf_p0"] = _np(p0).reshape(-1)
    tensors["rbf_p1"] = _np(p1).reshape(-1) if p1 is not None else np.zeros(int(n_rbf), np.float32)
    if kind_name == "SchNet":
        kind, nf, eps = KIND_SCHNET, int(rep.n_filters), 0.0
        L = len(rep.interactions)
        for l, it in enumerate(rep.interactions):
            ts = dict(in2f_w=it.in2f.weight, fn_w1=it.filter_network[0].weight, fn_b1=it.filter_network[0].bias,
                      fn_w2=it.filter_network[1].weight, fn_b2=it.filter_network[1].bias,
                      f2out_w1=it.f2out[0].weight, f2out_b1=it.f2out[0].bias,
                      f2out_w2=it.f2out[1].weight, f2out_b2=it.f2out[1].bias)
            for k, v in ts.items():
                tensors["l%d.%s" % (l, k)] = _np(v)
    else:
        kind, nf, eps = KIND_PAINN, F, float(rep._eps())
        L = int(rep.n_interactions)
        fw, fb = _np(rep.filter_net.weight), _np(rep.filter_net.bias)
        if rep.share_filters:                      # one [3F, n_rbf] block used by every interaction
            fw, fb = np.tile(fw, (L, 1)), np.tile(fb, L)
        tensors["filt_w"], tensors["filt_b"] = fw, fb
        for l in range(L):
            it, mx = rep.interactions[l], rep.mixing[l]
            ts = dict(ctx_w1=it.interatomic_context_net[0].weight, ctx_b1=it.interatomic_context_net[0].bias,
                      ctx_w2=it.interatomic_context_net[1].weight, ctx_b2=it.interatomic_context_net[1].bias,
                      mix_w=mx.mu_channel_mix.weight,
                      ictx_w1=mx.intraatomic_context_net[0].weight, ictx_b1=mx.intraatomic_context_net[0].bias,
                      ictx_w2=mx.intraatomic_context_net[1].weight, ictx_b2=mx.intraatomic_context_net[1].bias)
            for k, v in ts.items():
                tensors["l%d.%s" % (l, k)] = _np(v)
    H = int(net[0].out_features)
    tensors["head_w1"] = _np(net[0].weight)
    tensors["head_b1"] = _np(net[0].bias) if net[0].bias is not None else np.zeros(H, np.float32)
    tensors["head_w2"] = _np(net[1].weight).reshape(-1)
    tensors["head_b2"] = _np(net[1].bias).reshape(-1) if net[1].bias is not None else np.zeros(1, np.float32)
    mean, ext, atomref = _offsets_of(model, head.output_key)
    n_atomref = 0
    if atomref is not None:
        tensors["atomref"] = atomref.astype(np.float32)
        n_atomref = int(atomref.shape[0])
    names = list(tensors)
    ints = [VERSION, kind, F, nf, L, int(n_rbf), int(rbk), H, int(head_act), int(tensors["embedding"].shape[0]),
            ext, n_atomref, len(names), 0, 0, 0]
    head_bytes = MAGIC + struct.pack("<16i", *ints) + struct.pack("<4f", float(cutoff), eps, mean, 0.0)
    table, off = b"", 0
    for n in names:
        a = tensors[n]
        if len(n.encode()) > 31:
            raise ValueError(n)
        table += n.encode().ljust(32, b"\0") + struct.pack("<qq", int(a.size), off)
        off += (int(a.size) + 15) // 16 * 16      # every tensor starts on a 64-byte boundary
    pre = head_bytes + table
    pre += b"\0" * ((-len(pre)) % 64)
    data = bytearray(off * 4)
    off = 0
    for n in names:
        a = tensors[n].astype("<f4").reshape(-1)
        data[off * 4: off * 4 + a.size * 4] = a.tobytes()
        off += (int(a.size) + 15) // 16 * 16
    blob = pre + bytes(data)
    if path is not None:
        with open(path, "wb") as f:
            f.write(blob)
    return blob


class DeployedPotential:
    """ctypes handle on ``spk_potential_*`` -- numpy arrays in and out, no torch involved in the call."""

    def __init__(self, path_or_bytes):
        from . import _lib
        self._L = _lib.lib()
        self._h = ctypes.c_void_p()
        if isinstance(path_or_bytes, (bytes, bytearray)):
            buf = bytes(path_or_bytes)
            rc = self._L.spk_potential_from_memory(buf, len(buf), ctypes.byref(self._h))
        else:
            rc = self._L.spk_potential_load(str(path_or_bytes).encode(), ctypes.byref(self._h))
        _lib.check(rc)
        info = (ctypes.c_int32 * 8)()
        rc_ = ctypes.c_float()
        _lib.check(self._L.spk_potential_info(self._h, info, ctypes.byref(rc_)))
        self.info = dict(zip(["kind", "n_atom_basis", "n_interactions", "n_rbf", "radial", "n_filters", "head_hidden",
                              "embedding_rows"], list(info)))
        self.cutoff = float(rc_.value)

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._L.spk_potential_free(h)
        self._h = None

    __del__ = close

    @staticmethod
    def _p(a, ct):
        return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None

    def compute(self, z, R, idx_i, idx_j, offsets=None, idx_m=None, n_mol=1):
        from . import _lib
        z = np.ascontiguousarray(z, np.int64)
        R = np.ascontiguousarray(R, np.float32)
        ii = np.ascontiguousarray(idx_i, np.int64)
        jj = np.ascontiguousarray(idx_j, np.int64)
        off = np.ascontiguousarray(offsets, np.float32) if offsets is not None else None
        im = np.ascontiguousarray(idx_m, np.int64) if idx_m is not None else None
        n = int(z.shape[0])
        E = np.empty(int(n_mol), np.float32)
        Fo = np.empty((n, 3), np.float32)
        _lib.check(self._L.spk_potential_compute(
            self._h, n, self._p(z, ctypes.c_int64), self._p(R, ctypes.c_float), int(ii.shape[0]),
            self._p(ii, ctypes.c_int64), self._p(jj, ctypes.c_int64), self._p(off, ctypes.c_float), int(n_mol),
            self._p(im, ctypes.c_int64), self._p(E, ctypes.c_float), self._p(Fo, ctypes.c_float)))
        return E, Fo

    def compute_cell(self, z, R, cell=None, pbc=None, idx_m=None, n_mol=1, skin=0.0):
        from . import _lib
        z = np.ascontiguousarray(z, np.int64)
        R = np.ascontiguousarray(R, np.float32)
        c = np.ascontiguousarray(cell, np.float32).reshape(int(n_mol), 3, 3) if cell is not None else None
        pb = np.ascontiguousarray(pbc, np.uint8).reshape(int(n_mol), 3) if pbc is not None else None
        im = np.ascontiguousarray(idx_m, np.int64) if idx_m is not None else None
        n = int(z.shape[0])
        E = np.empty(int(n_mol), np.float32)
        Fo = np.empty((n, 3), np.float32)
        stats = (ctypes.c_int64 * 2)()
        _lib.check(self._L.spk_potential_compute_cell(
            self._h, n, self._p(z, ctypes.c_int64), self._p(R, ctypes.c_float), int(n_mol),
            self._p(im, ctypes.c_int64), self._p(c, ctypes.c_float), self._p(pb, ctypes.c_uint8), float(skin),
            self._p(E, ctypes.c_float), self._p(Fo, ctypes.c_float), stats))
        self.last_stats = {"pairs": int(stats[0]), "rebuilt": bool(stats[1])}
        return E, Fo


def main(argv=None):
    """``python -m schnetpack_amd.deploy model_path deployed_model_path`` -- the command line of
    src/scripts/spkdeploy:43-54.  ``model_path`` is a pickled ``NeuralNetworkPotential``; reference pickles
    resolve to the HIP-backed classes when ``schnetpack`` is importable (``schnetpack_amd.install``)."""
    import argparse
    import torch
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("model_path")
    ap.add_argument("deployed_model_path")
    args = ap.parse_args(argv)
    try:
        import schnetpack  # noqa: F401
        from . import install
        install.install()
    except ImportError:
        pass
    model = torch.load(args.model_path, map_location="cpu", weights_only=False)
    if not hasattr(model.representation, "electronic_embeddings"):   # utils/compatibility.py:36-39
        model.representation.electronic_embeddings = []
    blob = export_potential(model.eval(), args.deployed_model_path)
    print("stored deployed model at %s (%d bytes)." % (args.deployed_model_path, len(blob)))


if __name__ == "__main__":
    main()
