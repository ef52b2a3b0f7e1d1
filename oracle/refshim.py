"""Loader for the *real* reference hot-path modules (TEST INFRASTRUCTURE ONLY).

The reference package cannot be imported whole in this container (``ase``,
``pytorch_lightning``, ``hydra`` ... are absent), but every module on
the hot path depends on torch alone.  This shim registers path-only package stubs in
``sys.modules`` so that the individual sub-modules import unchanged from
``/root/reference/src``.  It is used only (a) by ``oracle/make_golden.py`` to generate the
committed fixtures under ``tests/golden`` and (b) by CPU tests that pin the oracle against
the live reference when ``/root/reference`` is present.  Nothing that runs on the GPU box
imports this file; ``/root/reference`` does not exist there.
"""
import importlib
import os
import sys
import types

REF_SRC = os.environ.get("SPK_REFERENCE_SRC", "/root/reference/src")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "schnetpack"))


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
