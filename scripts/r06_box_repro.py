"""Round 6: is the box-regime PaiNN force call bit-reproducible?  Energies / forces of repeated calls on the 10 125-atom box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, model as M, _lib
dev = torch.device("cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "painn"
b = S.water_box(n_side=15, seed=2)
rep = (O.init_painn_params if kind == "painn" else O.init_schnet_params)(); head = O.init_atomwise_params(128, seed=1)
m = M.build_model(kind); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
outs = []
for it in range(6):
    out = m(M.batch_to_inputs(b, dev))
    outs.append((out["energy"].detach().cpu().clone(), out["forces"].detach().cpu().clone()))
e0, f0 = outs[0]
for it, (e, f) in enumerate(outs[1:], 1):
    df = (f - f0).abs()
    print("call %d: energy equal %s (diff %.3e); forces: %d of %d atoms differ, max |diff| %.3e (max |F| %.3e)" % (
        it, torch.equal(e, e0), float((e - e0).abs().max()), int((df.max(1).values > 0).sum()), f.shape[0], float(df.max()), float(f0.abs().max())))
_lib.profile_enable(True); _lib.profile_report(); m(M.batch_to_inputs(b, dev)); torch.cuda.synchronize(); print(sorted(_lib.profile_report())); _lib.profile_enable(False)
