"""Round 6: glitch watch on the box-regime force calls (profiles/r06_box_split_glitch.md).  PaiNN: forces of N calls bit-identical (no atomics between
positions and forces).  SchNet: forces of N calls within 2e-6 of the first (float atomics in the backward: last-bit noise; a glitch was >= 1e-3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, model as M
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = S.water_box(n_side=15, seed=2)
for kind in ("painn", "schnet"):
    rep = (O.init_painn_params if kind == "painn" else O.init_schnet_params)(); head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
    inp = M.batch_to_inputs(b, dev)
    f0 = m(dict(inp))["forces"].detach().clone(); scale = float(f0.abs().max())
    worst, differ = 0.0, 0
    for it in range(n):
        f = m(dict(inp))["forces"].detach()
        dmax = float((f - f0).abs().max()) / scale
        worst = max(worst, dmax); differ += int(dmax > 0)
    print("%s: %d atoms, %d pairs, %d calls: %d differ from the first, worst |dF| / max |F| = %.3e" % (kind, b["Z"].shape[0], b["idx_i"].shape[0], n, differ, worst))
