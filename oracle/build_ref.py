"""Build recipe for ``oracle/_ref`` -- the reference itself as a travelling checker.  TEST INFRASTRUCTURE ONLY.

The reference is a Python package; "building" it means byte-compiling the package from the sources where
they lie under ``/root/reference/src/schnetpack`` into *sourceless* ``.pyc`` modules under
``oracle/_ref/src/schnetpack`` (same interpreter on the GPU box: same image), next to the two pretrained
model pickles the parity tests unpickle.  ``oracle/_ref/`` is git-ignored (no reference code enters the
history) but not gpurun-ignored, so -- like ``libspk_hip.so`` -- it travels to the GPU box, where
``/root/reference`` does not exist.  ``oracle/refshim.py`` imports the hot-path sub-modules from here when
``/root/reference`` is absent; consumers are the ``-m gpu`` tests (the reference's own ``NeuralNetworkPotential``
/ ``Atomwise`` / ``Forces`` on top of the HIP classes; the reference on the host CPU as the checker) and
``bench.py``'s ``cpu_baseline`` leg (``kind: "reference"``).  The product path never imports it.

    python -m oracle.build_ref [--force]

Called by ``__graft_entry__.build()`` whenever ``/root/reference`` is present.
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("SPK_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF_ROOT, "src", "schnetpack")
OUT = os.path.join(HERE, "_ref")
OUT_SRC = os.path.join(OUT, "src", "schnetpack")
# binary artefacts of the reference the tests need beside the code (pickled pretrained models + their geometry)
DATA = [
    ("interfaces/lammps/examples/aspirin/best_model", "data/lammps_aspirin_best_model"),
    ("interfaces/lammps/examples/aspirin/aspirin.data", "data/aspirin.data"),
    ("tests/testdata/md_ethanol.model", "data/md_ethanol.model"),
    ("tests/testdata/md_ethanol.xyz", "data/md_ethanol.xyz"),
] + [
    # the five PaiNN models trained on rMD17 ethanol that ship with the reference: real, TRAINED weights (a wider dynamic
    # range than any seeded initialisation) -- data artefacts of the reference, copied like the models above; the outputs of the reference on
    # them are committed as tests/golden/painn_rmd17_ethanol_trained.npz (oracle/make_golden.py)
    ("examples/trained_models/rmd17_ethanol/painn_%d/best_model" % k, "data/rmd17_ethanol_painn_%d.model" % k) for k in range(1, 6)
]
STAMP = os.path.join(OUT, "BUILD_INFO")


def source_available() -> bool:
    return os.path.isdir(SRC)


def built() -> bool:
    return os.path.exists(os.path.join(OUT_SRC, "properties.pyc"))


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every module of the reference package into ``oracle/_ref/src`` (sourceless byte code)."""
    if not source_available():
        if built():
            return OUT
        raise RuntimeError("reference sources not found at %s and oracle/_ref has not been built" % SRC)
    n = 0
    for dirpath, dirnames, filenames in os.walk(SRC):
        dirnames[:] = [d for d in dirnames if d != "__pycache__"]
        rel = os.path.relpath(dirpath, SRC)
        for fn in filenames:
            if not fn.endswith(".py"):
                continue
            src = os.path.join(dirpath, fn)
            dst = os.path.normpath(os.path.join(OUT_SRC, rel, fn + "c"))
            if not force and os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
                continue
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            # dfile: the name shown in tracebacks -- points back at the reference file
            py_compile.compile(src, cfile=dst, dfile=os.path.join("schnetpack", rel, fn), doraise=True, optimize=0)
            n += 1
    for rel_src, rel_dst in DATA:
        src = os.path.join(REF_ROOT, rel_src)
        dst = os.path.join(OUT, rel_dst)
        if os.path.exists(src) and (force or not os.path.exists(dst)):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
            n += 1
    with open(STAMP, "w") as fh:
        fh.write("byte-compiled from %s by oracle/build_ref.py with python %s\n" % (SRC, sys.version.split()[0]))
    if verbose:
        print("oracle/_ref: %d file(s) (re)built under %s" % (n, OUT))
    return OUT


def data_path(name: str) -> str:
    """Path of a reference data artefact: from /root/reference when present, else the copy under oracle/_ref."""
    for rel_src, rel_dst in DATA:
        if os.path.basename(rel_dst) == name:
            live = os.path.join(REF_ROOT, rel_src)
            return live if os.path.exists(live) else os.path.join(OUT, rel_dst)
    raise KeyError(name)




# ------------------------------------------------------------------------------------------------ deployed models
DEPLOYED = {"aspirin": "lammps_aspirin_best_model", "ethanol": "md_ethanol.model"}


def deployed_path(name: str) -> str:
    return os.path.join(OUT, "deployed", name + "_hip.pt")


def build_deployed(verbose: bool = True):
    """What `spkdeploy model deployed` (src/scripts/spkdeploy:16-40) produces when the HIP classes are installed: the
    shipped reference pickles are unpickled INTO the mirrors (``schnetpack_amd.install``), the reference's own
    ``NeuralNetworkPotential`` / ``Atomwise`` / ``Forces`` / ``AddOffsets`` code around them is scripted as is
    (``torch.jit.script``; casts dropped, ``AddOffsets.mean`` -> float32, ``cutoff`` metadata) and saved to
    ``oracle/_ref/deployed/<name>_hip.pt``.  Scripting needs the reference SOURCE text, so this runs in the build
    container only; the GPU tests ``torch.jit.load`` the archives the way interfaces/lammps/pair_schnetpack.cpp:128 does
    (after loading libspk_torch.so) and compare with the reference-generated fixtures tests/golden/deploy_painn.npz."""
    if not source_available():
        return []
    import numpy as np
    import torch
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import refshim
    import schnetpack_amd.install as inst
    if refshim.sourceless():
        return []
    refshim.load()
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    spk = sys.modules["schnetpack"]
    done = []
    inst.install(spk)
    try:
        load_model = sys.modules["schnetpack.utils"].load_model
        for name, data in DEPLOYED.items():
            dst = deployed_path(name)
            deps = [data_path(data), os.path.join(root, "schnetpack_amd", "csrc", "spk_torch.cpp")] + \
                   [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(root, "schnetpack_amd")) for f in fs if f.endswith(".py")]
            if os.path.exists(dst) and all(os.path.getmtime(dst) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
                done.append(dst)
                continue
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = load_model(data_path(data)).eval()
                keep = torch.nn.ModuleList()
                for pp in m.postprocessors:                 # spkdeploy:19-29
                    if type(pp).__name__ in ("CastTo64", "CastTo32"):
                        continue
                    if type(pp).__name__ == "AddOffsets":
                        pp.mean = pp.mean.float()
                    keep.append(pp)
                m.postprocessors = keep
                jm = torch.jit.script(m)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            meta = {"cutoff": str(jm.representation.cutoff.item()).encode("ascii")}      # spkdeploy:36
            torch.jit.save(jm, dst, _extra_files=meta)
            done.append(dst)
            if verbose:
                print("oracle/_ref: scripted %s -> %s" % (data, os.path.relpath(dst, root)))
    finally:
        inst.uninstall()
    return done


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_deployed()
