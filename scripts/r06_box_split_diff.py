"""Debug aid (round 6): forces of water boxes of growing size with the split path on / off (device vs device)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "schnet"
rep = O.init_schnet_params() if kind == "schnet" else O.init_painn_params()
head = O.init_atomwise_params(128, seed=1)
m = M.build_model(kind); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
for side in (4, 6, 8, 10, 12, 15):
    b = S.water_box(n_side=side, seed=3)
    res = {}
    for sp in (0, 1):
        _lib.set_split(sp)
        out = m(M.batch_to_inputs(b, dev))
        res[sp] = (out["energy"].detach().cpu().double(), out["forces"].detach().cpu().double())
    de = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max())
    df = (res[0][1] - res[1][1]).abs()
    print("side", side, "atoms", b["Z"].shape[0], "pairs", b["idx_i"].shape[0], "dE %.2e dF %.2e" % (de, float(df.max() / res[0][1].abs().max())),
          "atoms off by > 1e-4:", int((df.max(1).values > 1e-4 * res[0][1].abs().max()).sum()))
_lib.set_split(1)
