"""Round 6: are the two-launch force calls of the headline batches bit-reproducible?  (a glitch detector: profiles/r06_box_split_glitch.md)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, model as M, _lib
dev = torch.device("cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "schnet"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
b = S.molecule_batch("aspirin", 256, seed=11)
rep = (O.init_schnet_params if kind == "schnet" else O.init_painn_params)(); head = O.init_atomwise_params(128, seed=1)
m = M.build_model(kind); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
inp = M.batch_to_inputs(b, dev)
out0 = m(dict(inp)); f0 = out0["forces"].detach().clone(); e0 = out0["energy"].detach().clone()
scale = float(f0.abs().max())
bad_calls = 0
for it in range(n):
    out = m(dict(inp)); f = out["forces"].detach(); e = out["energy"].detach()
    df = (f - f0).abs().max(1).values
    nd = int((df > 0).sum())
    if nd or not torch.equal(e, e0):
        bad_calls += 1
        idx = torch.nonzero(df > 0).flatten()
        print("call %d: %d atoms differ, max |dF| / max |F| = %.3e, molecules %s; energies differ in %d molecules (max %.3e)" % (
            it, nd, float(df.max()) / scale, sorted(set((idx // 21).tolist()))[:8], int((e != e0).sum()), float((e - e0).abs().max())))
print(kind, "split", _lib.get_split(), ": %d of %d calls differ from the first" % (bad_calls, n))
