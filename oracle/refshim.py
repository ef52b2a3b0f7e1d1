"""Loader for the *real* reference hot-path modules (TEST INFRASTRUCTURE ONLY).

The reference package cannot be imported whole in this container (``ase``,
``pytorch_lightning``, ``hydra`` ... are absent), but every module on
the hot path depends on torch alone.  This shim registers path-only package stubs in
``sys.modules`` so that the individual sub-modules import unchanged from
``/root/reference/src`` -- or, where that does not exist (the GPU box), from the byte-compiled
build of the same package under ``oracle/_ref/src`` (``oracle/build_ref.py``; git-ignored, travels
with the snapshot like the built ``.so``).  Users: ``oracle/make_golden.py`` (fixtures under
``tests/golden``), the CPU tests that pin the oracle against the live reference, the ``-m gpu``
tests that run the reference's own callers on top of the HIP classes / use the reference on the
host CPU as the checker, and ``bench.py``'s ``cpu_baseline`` leg.  The product path never imports it.
"""
import importlib
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIVE = "/root/reference/src"
_BUILT = os.path.join(_HERE, "_ref", "src")
REF_SRC = os.environ.get("SPK_REFERENCE_SRC") or (_LIVE if os.path.isdir(os.path.join(_LIVE, "schnetpack")) else _BUILT)


def available() -> bool:
    root = os.path.join(REF_SRC, "schnetpack")
    return os.path.isdir(root) and (os.path.exists(os.path.join(root, "properties.py")) or os.path.exists(os.path.join(root, "properties.pyc")))


def sourceless() -> bool:
    """True when the reference is the byte-compiled build (no ``.py`` files: TorchScript of it is unavailable)."""
    return not os.path.exists(os.path.join(REF_SRC, "schnetpack", "properties.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Return a namespace with the reference classes of the hot path."""
    if not available():
        raise RuntimeError("reference sources not found at %s" % REF_SRC)
    if "schnetpack" in sys.modules and getattr(sys.modules["schnetpack"], "_spk_shim", False):
        return sys.modules["schnetpack"]._ns
    root = os.path.join(REF_SRC, "schnetpack")
    sys.dont_write_bytecode = True     # never drop __pycache__ directories into the reference tree
    import torch
    jit_script = torch.jit.script
    if sourceless():
        # nn/scatter.py:26 decorates a helper with @torch.jit.script at import time, which needs the source text;
        # the byte-compiled build runs that helper as plain Python -- the same ATen calls
        torch.jit.script = lambda fn=None, *a, **k: fn
    try:
        return _load(root)
    finally:
        torch.jit.script = jit_script


def _load(root):
    pkg = types.ModuleType("schnetpack")
    pkg.__path__ = [root]
    pkg.__version__ = "2.2.0"
    pkg._spk_shim = True
    sys.modules["schnetpack"] = pkg
    pkg.properties = importlib.import_module("schnetpack.properties")
    pkg.nn = importlib.import_module("schnetpack.nn")
    pkg.utils = importlib.import_module("schnetpack.utils")
    for sub in ("representation", "atomistic", "model", "transform", "data"):
        m = types.ModuleType("schnetpack." + sub)
        m.__path__ = [os.path.join(root, sub)]
        sys.modules["schnetpack." + sub] = m
        setattr(pkg, sub, m)
    # third-party modules imported at the top of transform/neighborlist.py; unused by the
    # pure-torch neighbour list that we need
    for name in ("fasteners", "ase", "ase.neighborlist", "ase.data", "matscipy",
                 "matscipy.neighbours", "dirsync", "vesin"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["ase"].Atoms = object
    sys.modules["ase.neighborlist"].neighbor_list = None
    sys.modules["matscipy.neighbours"].neighbour_list = None
    sys.modules["dirsync"].sync = None
    sys.modules["vesin"].NeighborList = object
    ns = types.SimpleNamespace()
    tb = importlib.import_module("schnetpack.transform.base")
    sys.modules["schnetpack.transform"].Transform = tb.Transform
    ns.schnet = importlib.import_module("schnetpack.representation.schnet")
    ns.painn = importlib.import_module("schnetpack.representation.painn")
    ns.distances = importlib.import_module("schnetpack.atomistic.distances")
    ns.atomwise = importlib.import_module("schnetpack.atomistic.atomwise")
    ns.response = importlib.import_module("schnetpack.atomistic.response")
    ns.model = importlib.import_module("schnetpack.model.base")
    ns.loader = importlib.import_module("schnetpack.data.loader")
    try:
        ns.neighborlist = importlib.import_module("schnetpack.transform.neighborlist")
    except Exception as exc:  # pragma: no cover - diagnostic only
        ns.neighborlist = None
        ns.neighborlist_error = exc
    ns.nn = pkg.nn
    ns.properties = pkg.properties
    pkg._ns = ns
    return ns
