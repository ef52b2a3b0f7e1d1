"""Round 6: achieved error of the molecule-resident kernels with the split-precision matrix path on / off, against the float64
oracle (and the float32 oracle = the reference's arithmetic) on the batches of tests/test_gpu_mol.py / test_gpu_painn_mol.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def run(kind, b, n_int, n_rbf, radial, split):
    rep = (O.init_schnet_params if kind == "schnet" else O.init_painn_params)(128, n_int, n_rbf, 5.0, radial=radial)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind, 128, n_int, n_rbf, 5.0, radial)
    M.load_reference_params(m, rep, head)
    m = m.to("cuda").eval()
    _lib.set_split(split)
    inp = M.batch_to_inputs(b, torch.device("cuda"))
    out = m(inp)
    return (out["energy"].detach().cpu(), out["forces"].detach().cpu(), inp["scalar_representation"].detach().cpu()), rep, head


kinds = [k for k in sys.argv[1:] if k in ("schnet", "painn")] if len(sys.argv) > 1 else ["schnet"]
for kind in kinds:
    for name, b, n_int, n_rbf, radial in [
        ("aspirin x 7", S.molecule_batch("aspirin", 7, seed=3), 3, 20, "gaussian"),
        ("aspirin x 64", S.molecule_batch("aspirin", 64, seed=9), 3, 20, "gaussian"),
        ("aspirin x 16 bessel", S.molecule_batch("aspirin", 16, seed=5), 3, 20, "bessel"),
        ("aspirin x 8, 6 interactions bessel", S.molecule_batch("aspirin", 8, seed=7), 6, 20, "bessel"),
    ]:
        torch.manual_seed(0)
        res = {}
        for split in (0, 1):
            torch.manual_seed(0)
            res[split], rep, head = run(kind, b, n_int, n_rbf, radial, split)
        r64 = O.energy_and_forces(kind, rep, head, b, n_int, dtype=torch.float64, need_rep=True)
        r32 = O.energy_and_forces(kind, rep, head, b, n_int, dtype=torch.float32, need_rep=True)
        print("%s, %s:" % (kind, name))
        print("   float32 oracle vs float64: x %.2e  E %.2e  F %.2e" % (rel(r32["scalar_representation"], r64["scalar_representation"]), rel(r32["energy"], r64["energy"]), rel(r32["forces"], r64["forces"])))
        for split in (0, 1):
            e, f, x = res[split]
            print("   split=%d vs float64: x %.2e  E %.2e  F %.2e   | vs float32 oracle: x %.2e  E %.2e  F %.2e" % (
                split, rel(x, r64["scalar_representation"]), rel(e, r64["energy"]), rel(f, r64["forces"]),
                rel(x, r32["scalar_representation"]), rel(e, r32["energy"]), rel(f, r32["forces"])))
_lib.set_split(1)
if os.environ.get("PROBE_WHERE"):
    b = S.molecule_batch("aspirin", 7, seed=3)
    (e, f, x), rep, head = run("schnet", b, 3, 20, "gaussian", 1)
    r64 = O.energy_and_forces("schnet", rep, head, b, 3, dtype=torch.float64, need_rep=True)
    err = (x.double() - r64["scalar_representation"]).abs() / r64["scalar_representation"].abs().max()
    print("entries > 2e-6:", int((err > 2e-6).sum()), "of", err.numel())
    idx = torch.nonzero(err > 2e-6)
    print("atoms:", sorted(set(idx[:, 0].tolist()))[:40])
    print("channels:", sorted(set(idx[:, 1].tolist()))[:64])
    for l in (1, 2):
        (e, f, x), rep, head = run("schnet", b, l, 20, "gaussian", 1)
        r = O.energy_and_forces("schnet", rep, head, b, l, dtype=torch.float64, need_rep=True)
        err = (x.double() - r["scalar_representation"]).abs() / r["scalar_representation"].abs().max()
        print("n_int", l, "max err", float(err.max()), "entries > 2e-6:", int((err > 2e-6).sum()))
