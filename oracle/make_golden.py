"""Generate the committed fixtures under tests/golden from the LIVE reference
(``/root/reference`` imported through ``oracle/refshim.py``).  TEST INFRASTRUCTURE ONLY.

Run in the build container (the GPU box has no reference):

    python oracle/make_golden.py

Every fixture stores the inputs, the reference outputs and (for seeded-init models) a
checksum of the weights; the weights themselves are reproduced on any box by
``spk_oracle.init_*_params`` (same torch CPU RNG stream as the reference constructors).
The one real-weight fixture (the shipped PaiNN aspirin model,
``interfaces/lammps/examples/aspirin/best_model``) stores its state dict.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim, spk_oracle as O  # noqa: E402
from schnetpack_amd import synthetic as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref_inputs(b):
    n_mol = int(b["n_mol"])
    counts = torch.bincount(b["idx_m"], minlength=n_mol)
    return {
        "_atomic_numbers": b["Z"], "_positions": b["R"].clone(), "_idx_i": b["idx_i"],
        "_idx_j": b["idx_j"], "_offsets": b["offsets"], "_idx_m": b["idx_m"],
        "_cell": torch.zeros(n_mol, 3, 3), "_pbc": torch.zeros(3 * n_mol, dtype=torch.bool),
        "_n_atoms": counts,
    }


def checksum(p):
    return float(sum(v.double().abs().sum() for v in p.values()))


def run_reference(ns, rep, head_sd, b):
    aw = ns.atomwise.Atomwise(n_in=rep.n_atom_basis, output_key="energy")
    aw.load_state_dict(head_sd)
    model = ns.model.NeuralNetworkPotential(
        rep, input_modules=[ns.distances.PairwiseDistances()],
        output_modules=[aw, ns.response.Forces()])
    model.eval()
    inp = ref_inputs(b)
    out = model(inp)
    res = {"energy": out["energy"].detach().numpy(), "forces": out["forces"].detach().numpy(),
           "scalar_representation": inp["scalar_representation"].detach().numpy()}
    if "vector_representation" in inp:
        res["vector_representation"] = inp["vector_representation"].detach().numpy()
    return res


def save(name, b, res, **extra):
    arrs = {"in_" + k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in b.items() if k != "cell"}
    arrs.update({"ref_" + k: v for k, v in res.items()})
    arrs.update(extra)
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in arrs.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = refshim.load()
    nn = ns.nn

    # --- L0 known answers (the reference's own golden tests, tests/nn/test_radial.py:6-74,
    #     test_cutoff.py:7-42, test_activations.py:7-25, evaluated by the reference modules)
    d = torch.linspace(0.0, 6.0, 25)
    d2 = torch.tensor([[0.0, 0.3], [1.7, 4.99], [5.0, 7.5]])
    ka = {
        "d": d.numpy(), "d2": d2.numpy(),
        "gauss20_5": nn.GaussianRBF(20, 5.0)(d).numpy(),
        "gauss5_1p5_start0p5": nn.GaussianRBF(5, 1.5, start=0.5)(d2).numpy(),
        "bessel20_5": nn.BesselRBF(20, 5.0)(d).numpy(),
        "bessel7_3": nn.BesselRBF(7, 3.0)(d2).numpy(),
        "cos5": nn.CosineCutoff(5.0)(d).numpy(),
        "cos1p8": nn.CosineCutoff(1.8)(d2).numpy(),
        "ssp_x": torch.linspace(-30, 30, 61).numpy(),
        "ssp_y": nn.shifted_softplus(torch.linspace(-30, 30, 61)).numpy(),
    }
    # scatter_add golden incl. trailing dims, dim=1, unsorted indices (nn/scatter.py:7-34)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(17, 3, 8, generator=g)
    idx = torch.randint(0, 6, (17,), generator=g)
    ka["scat_x"] = x.numpy(); ka["scat_idx"] = idx.numpy()
    ka["scat_y0"] = nn.scatter_add(x, idx, dim_size=7).numpy()
    xt = x.permute(1, 0, 2).contiguous()
    ka["scat_y1"] = nn.scatter_add(xt, idx, dim_size=7, dim=1).numpy()
    np.savez_compressed(os.path.join(OUT, "nn_known_answers.npz"), **ka)
    print("wrote nn_known_answers.npz")

    head = O.init_atomwise_params(128, seed=1)

    # --- cfg 1: ethanol, SchNet(128,3), seed 0
    cases = [
        ("schnet_ethanol", "schnet", S.molecule_batch("ethanol", 1, jitter=0.0), dict()),
        ("schnet_aspirin8", "schnet", S.molecule_batch("aspirin", 8, seed=0), dict()),
        ("painn_ethanol", "painn", S.molecule_batch("ethanol", 1, jitter=0.0), dict()),
        ("painn_aspirin8", "painn", S.molecule_batch("aspirin", 8, seed=0), dict()),
        ("schnet_bessel_aspirin2", "schnet", S.molecule_batch("aspirin", 2, seed=3),
         dict(radial="bessel")),
        ("painn_bessel_aspirin2", "painn", S.molecule_batch("aspirin", 2, seed=3),
         dict(radial="bessel")),
        # short cutoff: some pairs beyond the cutoff are kept in the list (skin-style) so that
        # the [d < rc] mask of the cosine cutoff is exercised
        ("schnet_skin_aspirin2", "schnet", S.molecule_batch("aspirin", 2, cutoff=5.0, seed=4),
         dict(cutoff=3.5)),
        ("painn_skin_aspirin2", "painn", S.molecule_batch("aspirin", 2, cutoff=5.0, seed=4),
         dict(cutoff=3.5)),
    ]
    # periodic boundary conditions: 64 water molecules in a 12.4 A box (cell offsets, ~54 neighbours/atom)
    wb = S.water_box(n_side=4, seed=0)
    cases += [("schnet_water192", "schnet", wb, dict()), ("painn_water192", "painn", wb, dict())]
    for name, kind, b, kw in cases:
        cutoff = kw.get("cutoff", 5.0)
        radial = kw.get("radial", "gaussian")
        rb = nn.GaussianRBF(20, cutoff) if radial == "gaussian" else nn.BesselRBF(20, cutoff)
        torch.manual_seed(0)
        if kind == "schnet":
            rep = ns.schnet.SchNet(128, 3, rb, nn.CosineCutoff(cutoff))
            p = O.init_schnet_params(cutoff=cutoff, radial=radial)
        else:
            rep = ns.painn.PaiNN(128, 3, rb, nn.CosineCutoff(cutoff))
            p = O.init_painn_params(cutoff=cutoff, radial=radial)
        sd = rep.state_dict()
        assert all(torch.equal(sd[k], p[k].to(sd[k].dtype)) for k in sd), name
        res = run_reference(ns, rep, head, b)
        save(name + ".npz", b, res, weights_checksum=checksum(p), kind=kind, cutoff=cutoff,
             radial=radial, n_interactions=3)

    # --- real weights: shipped PaiNN aspirin model (2 interactions)
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    m = torch.load(os.path.join(refshim.REF_SRC, "..", "interfaces", "lammps", "examples",
                                "aspirin", "best_model"), map_location="cpu", weights_only=False)
    m.eval()
    rep = m.representation
    head_sd = m.output_modules[0].state_dict()
    b = S.molecule_batch("aspirin", 4, seed=7, jitter=0.03)
    res = run_reference(ns, rep, head_sd, b)
    w = {"w_rep." + k: v.numpy() for k, v in rep.state_dict().items()}
    w.update({"w_head." + k: v.numpy() for k, v in head_sd.items()})
    save("painn_aspirin_pretrained.npz", b, res, kind="painn", cutoff=float(rep.cutoff),
         radial="gaussian", n_interactions=int(rep.n_interactions), **w)


def trained_model_goldens(ns):
    """The reference's five trained rMD17-ethanol PaiNN models (examples/trained_models/rmd17_ethanol/painn_{1..5}/best_model: PaiNN(128, 3,
    20 Gaussians, 5 A) + Atomwise + Forces -- configs[3]'s architecture with TRAINED weights): representation + energy head (no
    postprocessors, like every other fixture) on six jittered ethanol frames.  The weights are not stored (12 MB): the tests load them from
    the same model files (oracle/build_ref.py copies them into oracle/_ref/data) and check them against the checksum stored here."""
    from oracle import build_ref
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    b = S.molecule_batch("ethanol", 6, seed=17, jitter=0.06)
    arrs = {"in_" + k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in b.items() if k != "cell"}
    for k in range(1, 6):
        m = torch.load(build_ref.data_path("rmd17_ethanol_painn_%d.model" % k), map_location="cpu", weights_only=False)
        m.eval()
        rep, head_sd = m.representation, m.output_modules[0].state_dict()
        res = run_reference(ns, rep, head_sd, b)
        for q, v in res.items():
            arrs["ref%d_%s" % (k, q)] = v
        arrs["weights_checksum_%d" % k] = checksum({**rep.state_dict(), **{"head." + kk: v for kk, v in head_sd.items()}})
        w = torch.cat([v.flatten().double() for v in rep.state_dict().values() if v.is_floating_point()])
        arrs["weights_absmax_%d" % k] = float(w.abs().max())
    arrs.update(kind="painn", cutoff=5.0, radial="gaussian", n_interactions=3, n_models=5)
    np.savez_compressed(os.path.join(OUT, "painn_rmd17_ethanol_trained.npz"), **arrs)
    print("wrote painn_rmd17_ethanol_trained.npz", {k: getattr(v, "shape", v) for k, v in arrs.items() if not k.startswith("ref")})


def neighbor_list_goldens(ns):
    """(1) the reference's own precomputed Argon vectors (tests/conftest.py:192-447), lifted out of the
    fixture functions; (2) TorchNeighborList outputs (transform/neighborlist.py:438-553) on seeded
    systems: orthorhombic, triclinic + mixed pbc, a cell smaller than the cutoff (several images of the
    same atom), no pbc, fp64 positions."""
    import ast
    import types
    from oracle import nbl_oracle as NB
    src = open(os.path.join(refshim.REF_SRC, "..", "tests", "conftest.py")).read()
    props = types.SimpleNamespace(Z="_atomic_numbers", R="_positions", cell="_cell", pbc="_pbc", n_atoms="_n_atoms",
                                  idx_i="_idx_i", idx_j="_idx_j", offsets="_offsets", Rij="_Rij")
    env = {"np": np, "torch": torch, "spk": types.SimpleNamespace(properties=props)}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("environment_periodic", "environment_nonperiodic"):
            node.decorator_list = []
            exec(compile(ast.Module([node], []), "conftest", "exec"), env)
    arrs = {}
    for tag in ("periodic", "nonperiodic"):
        cutoff, p, nb = env["environment_" + tag]()
        arrs[tag + "_cutoff"] = cutoff
        for k in ("_positions", "_cell", "_pbc"):
            arrs[tag + k] = p[k].numpy()
        for k, v in nb.items():
            arrs[tag + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "nbl_argon.npz"), **arrs)
    print("wrote nbl_argon.npz", {k: getattr(v, "shape", v) for k, v in arrs.items()})

    g = torch.Generator().manual_seed(11)
    cases = {
        "ortho": (torch.rand(60, 3, generator=g) * torch.tensor([7.0, 6.0, 8.0]), torch.diag(torch.tensor([7.0, 6.0, 8.0])), [True, True, True], 5.0),
        "triclinic_mixed": (torch.rand(50, 3, generator=g) * 9.0 - 1.0, torch.tensor([[9.0, 0.0, 0.0], [2.5, 8.0, 0.0], [1.0, -1.5, 10.0]]), [True, True, False], 4.0),
        "small_cell": (torch.rand(12, 3, generator=g) * 3.5, torch.tensor([[3.6, 0.0, 0.0], [0.4, 3.5, 0.0], [0.0, 0.3, 4.0]]), [True, True, True], 5.0),
        "free": (torch.randn(80, 3, generator=g) * 4.0, torch.zeros(3, 3), [False, False, False], 3.0),
        # (positions inside the cell along the periodic axes: TorchNeighborList only searches +-ceil(cutoff/height)
        #  images of the UNWRAPPED positions, so it misses pairs of atoms that sit several cells apart)
        "slab_fp64": (torch.rand(40, 3, generator=g, dtype=torch.float64) * torch.tensor([6.0, 6.5, 12.0], dtype=torch.float64), torch.diag(torch.tensor([6.0, 6.5, 30.0], dtype=torch.float64)), [True, True, False], 4.5),
    }
    arrs = {"names": np.array(sorted(cases))}
    for name, (R, cell, pbc, rc) in cases.items():
        pbc = torch.tensor(pbc)
        nl = ns.neighborlist.TorchNeighborList(rc)
        i, j, off = nl._build_neighbor_list(None, R, cell, pbc, rc)
        S = torch.round(off @ torch.linalg.inv(cell)).long() if bool(pbc.any()) else torch.zeros(i.shape[0], 3, dtype=torch.long)
        order = NB.canonical_order(i, j, S)
        arrs.update({name + "_R": R.numpy(), name + "_cell": cell.numpy(), name + "_pbc": pbc.numpy(), name + "_cutoff": rc,
                     name + "_idx_i": i[order].numpy(), name + "_idx_j": j[order].numpy(), name + "_S": S[order].numpy(),
                     name + "_offsets": off[order].numpy()})
        print("  nbl case", name, "pairs", int(i.shape[0]))
    np.savez_compressed(os.path.join(OUT, "nbl_cases.npz"), **arrs)
    print("wrote nbl_cases.npz")


def ring_polymer_goldens():
    """Execute the reference's own RingPolymer._init_propagator / _main_step (md/integrators.py:152-229)
    and NormalModeTransformer (md/utils/normal_model_transformation.py) on seeded beads.  The two modules
    pull in the whole MD package, so the two methods are lifted out with ``ast`` and run against a
    minimal System stand-in that has exactly the normal-mode properties of md/system.py:444-482."""
    import ast
    import importlib.util
    import types
    md_dir = os.path.join(refshim.REF_SRC, "schnetpack", "md")
    spec = importlib.util.spec_from_file_location("_ref_nmt", os.path.join(md_dir, "utils", "normal_model_transformation.py"))
    nmt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nmt)
    tree = ast.parse(open(os.path.join(md_dir, "integrators.py")).read())
    rp = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RingPolymer"][0]
    fns = {}
    env = {"torch": torch, "np": np, "System": object}
    for node in rp.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("_init_propagator", "_main_step"):
            exec(compile(ast.Module([node], []), "integrators", "exec"), env)
            fns[node.name] = env[node.name]

    class Sys:
        def __init__(self, q, p, m, nb):
            self.positions, self.momenta, self.masses = q, p, m
            self.nm_transform = nmt.NormalModeTransformer(nb)
        positions_normal = property(lambda s: s.nm_transform.beads2normal(s.positions),
                                    lambda s, v: setattr(s, "positions", s.nm_transform.normal2beads(v)))
        momenta_normal = property(lambda s: s.nm_transform.beads2normal(s.momenta),
                                  lambda s, v: setattr(s, "momenta", s.nm_transform.normal2beads(v)))

    arrs = {}
    g = torch.Generator().manual_seed(21)
    for nb in (1, 2, 4, 5, 8):
        omega, dt = 40.0 + 3.0 * nb, 0.0005
        me = types.SimpleNamespace(n_beads=nb, omega=omega, time_step=dt)
        omega_normal, prop = fns["_init_propagator"](me)
        me.propagator = prop
        q = torch.randn(nb, 7, 3, generator=g, dtype=torch.float64)
        p = torch.randn(nb, 7, 3, generator=g, dtype=torch.float64)
        m = (torch.rand(1, 7, 1, generator=g, dtype=torch.float64) * 15 + 1)
        sysm = Sys(q.clone(), p.clone(), m, nb)
        fns["_main_step"](me, sysm)
        t = "b%d_" % nb
        arrs.update({t + "omega": omega, t + "dt": dt, t + "C": sysm.nm_transform.c_transform.numpy(),
                     t + "propagator": prop[..., 0, 0].numpy(), t + "omega_normal": omega_normal.numpy(),
                     t + "q": q.numpy(), t + "p": p.numpy(), t + "m": m.numpy(),
                     t + "q_out": sysm.positions.numpy(), t + "p_out": sysm.momenta.numpy()})
    np.savez_compressed(os.path.join(OUT, "md_ring_polymer.npz"), **arrs)
    print("wrote md_ring_polymer.npz", sorted(k for k in arrs if k.startswith("b8_")))


def deploy_goldens(ns):
    """Deployed-model goldens: the shipped PaiNN models processed like
    src/scripts/spkdeploy:16-31 does (dtype casts dropped, AddOffsets.mean -> float32, torch.jit.script), fed
    the input dict interfaces/lammps/pair_schnetpack.cpp:285-301 builds (one system, idx_m = 0, edges in
    neighbour-list order of the LOCAL atom index -- here a random permutation -- and offsets = image shifts).
    Two geometries per model: the free molecule and the same molecule in a small periodic cell (images inside
    the cutoff).  Weights + AddOffsets statistics are stored so that the GPU box can rebuild the model."""
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    casts = ("CastTo64", "CastTo32")
    models = {
        "aspirin": os.path.join(refshim.REF_SRC, "..", "interfaces", "lammps", "examples", "aspirin", "best_model"),
        "ethanol": os.path.join(refshim.REF_SRC, "..", "tests", "testdata", "md_ethanol.model"),
    }
    arrs = {}
    g = torch.Generator().manual_seed(21)
    for name, path in models.items():
        m = torch.load(path, map_location="cpu", weights_only=False)
        if not hasattr(m.representation, "electronic_embeddings"):    # utils/compatibility.py:36-39 (2.0.4 pickles)
            m.representation.electronic_embeddings = []
        m.eval()
        keep = torch.nn.ModuleList()
        for pp in m.postprocessors:                 # spkdeploy:19-29
            if type(pp).__name__ in casts:
                continue
            if type(pp).__name__ == "AddOffsets":
                pp.mean = pp.mean.float()
            keep.append(pp)
        m.postprocessors = keep
        try:
            jm = torch.jit.script(m)
            scripted = True
        except Exception as e:  # pragma: no cover - the eager module computes the same function
            print("  torch.jit.script failed (%s); using the eager module" % type(e).__name__)
            jm, scripted = m, False
        cutoff = float(m.representation.cutoff.item())
        b = S.molecule_batch(name, 1, seed=5, jitter=0.02, cutoff=cutoff)
        Z, R0 = b["Z"], b["R"].float()
        n = int(Z.shape[0])
        cell = torch.tensor([[7.5, 0.0, 0.0], [0.6, 8.0, 0.0], [0.0, -0.4, 7.0]])
        for tag, pbc in (("free", torch.zeros(3, dtype=torch.bool)), ("pbc", torch.ones(3, dtype=torch.bool))):
            R = R0 - R0.min(0).values + 0.3 if tag == "pbc" else R0
            nl = ns.neighborlist.TorchNeighborList(cutoff)
            i, j, off = nl._build_neighbor_list(Z, R, cell if tag == "pbc" else torch.zeros(3, 3), pbc, cutoff)
            perm = torch.randperm(int(i.shape[0]), generator=g)
            i, j, off = i[perm].contiguous(), j[perm].contiguous(), off[perm].float().contiguous()
            inp = {"_positions": R.clone(), "_idx_i": i, "_idx_j": j, "_idx_m": torch.zeros(n, dtype=torch.long), "_offsets": off,
                   "_cell": (cell if tag == "pbc" else torch.zeros(3, 3)), "_n_atoms": torch.tensor([n]), "_atomic_numbers": Z}
            out = jm(inp)
            t = "%s_%s_" % (name, tag)
            arrs.update({t + "Z": Z.numpy(), t + "R": R.numpy(), t + "idx_i": i.numpy(), t + "idx_j": j.numpy(), t + "offsets": off.numpy(),
                         t + "cell": inp["_cell"].numpy(), t + "pbc": pbc.numpy(), t + "energy": out["energy"].detach().float().numpy(),
                         t + "forces": out["forces"].detach().float().numpy()})
            print("  deploy case", t, "edges", int(i.shape[0]), "E", float(out["energy"]), "scripted", scripted)
        rep = m.representation
        arrs[name + "_cutoff"] = cutoff
        arrs[name + "_n_interactions"] = int(rep.n_interactions)
        arrs[name + "_mean"] = float(keep[0].mean) if len(keep) else 0.0
        arrs[name + "_scripted"] = scripted
        if name == "aspirin":       # these weights are already in painn_aspirin_pretrained.npz (w_rep.* / w_head.*)
            arrs[name + "_weights_checksum"] = checksum(rep.state_dict()) + checksum(m.output_modules[0].state_dict())
            continue
        for k, v in rep.state_dict().items():
            arrs[name + "_w_rep." + k] = v.numpy()
        for k, v in m.output_modules[0].state_dict().items():
            arrs[name + "_w_head." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "deploy_painn.npz"), **arrs)
    print("wrote deploy_painn.npz")


def deep_model_goldens(ns):
    """The reference's DEFAULT depth (configs/model/representation/schnet.yaml:5-9: n_interactions = 6; PaiNN with six blocks
    beside it): twice the interactions of the bench configuration -- twice the error accumulation of the fp32 kernels (round-2
    review: pin a 6-interaction golden).  Molecule batch and the periodic 192-atom water box."""
    nn = ns.nn
    head = O.init_atomwise_params(128, seed=1)
    wb = S.water_box(n_side=4, seed=0)
    for name, kind, b in (("schnet6_aspirin4", "schnet", S.molecule_batch("aspirin", 4, seed=11)), ("schnet6_water192", "schnet", wb),
                          ("painn6_aspirin4", "painn", S.molecule_batch("aspirin", 4, seed=11)), ("painn6_water192", "painn", wb)):
        rb = nn.GaussianRBF(20, 5.0)
        torch.manual_seed(0)
        if kind == "schnet":
            rep = ns.schnet.SchNet(128, 6, rb, nn.CosineCutoff(5.0))
            p = O.init_schnet_params(n_interactions=6)
        else:
            rep = ns.painn.PaiNN(128, 6, rb, nn.CosineCutoff(5.0))
            p = O.init_painn_params(n_interactions=6)
        sd = rep.state_dict()
        assert all(torch.equal(sd[k], p[k].to(sd[k].dtype)) for k in sd), name
        res = run_reference(ns, rep, head, b)
        save(name + ".npz", b, res, weights_checksum=checksum(p), kind=kind, cutoff=5.0, radial="gaussian", n_interactions=6)


def deep_bessel_goldens(ns):
    """Six interactions x BesselRBF (round-3 review): the molecule kernels evaluate sin / cos with the hardware transcendentals, and
    depth multiplies whatever error they leave -- the least-margin combination gets its own reference fixtures, on a molecule batch
    and on the periodic 192-atom water box."""
    nn = ns.nn
    head = O.init_atomwise_params(128, seed=1)
    wb = S.water_box(n_side=4, seed=0)
    for name, kind, b in (("schnet6_bessel_aspirin4", "schnet", S.molecule_batch("aspirin", 4, seed=11)), ("schnet6_bessel_water192", "schnet", wb),
                          ("painn6_bessel_aspirin4", "painn", S.molecule_batch("aspirin", 4, seed=11)), ("painn6_bessel_water192", "painn", wb)):
        rb = nn.BesselRBF(20, 5.0)
        torch.manual_seed(0)
        if kind == "schnet":
            rep = ns.schnet.SchNet(128, 6, rb, nn.CosineCutoff(5.0))
            p = O.init_schnet_params(n_interactions=6, radial="bessel")
        else:
            rep = ns.painn.PaiNN(128, 6, rb, nn.CosineCutoff(5.0))
            p = O.init_painn_params(n_interactions=6, radial="bessel")
        sd = rep.state_dict()
        assert all(torch.equal(sd[k], p[k].to(sd[k].dtype)) for k in sd), name
        res = run_reference(ns, rep, head, b)
        save(name + ".npz", b, res, weights_checksum=checksum(p), kind=kind, cutoff=5.0, radial="bessel", n_interactions=6)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deep":
        deep_model_goldens(refshim.load())
    elif len(sys.argv) > 1 and sys.argv[1] == "deep_bessel":
        deep_bessel_goldens(refshim.load())
    elif len(sys.argv) > 1 and sys.argv[1] == "md":
        ring_polymer_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "nbl":
        neighbor_list_goldens(refshim.load())
    elif len(sys.argv) > 1 and sys.argv[1] == "deploy":
        deploy_goldens(refshim.load())
    elif len(sys.argv) > 1 and sys.argv[1] == "trained":
        trained_model_goldens(refshim.load())
    else:
        main()
        neighbor_list_goldens(refshim.load())
        ring_polymer_goldens()
        deploy_goldens(refshim.load())
        deep_model_goldens(refshim.load())
        deep_bessel_goldens(refshim.load())
        trained_model_goldens(refshim.load())
