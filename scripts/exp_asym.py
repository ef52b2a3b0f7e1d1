"""Round 4: per-kernel times of the SchNet representation forward + backward on the padded-neighbour sweep graphs (symmetric ring
graph vs random directed graph, N = 16 384, k = 32) -- where the 4.7x cliff of asymmetric lists comes from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from schnetpack_amd import _lib, model as M, synthetic as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_model("schnet").to(dev).eval()
rep = m.representation
for name, b in (("symmetric", S.ring_graph_batch(16384, 32, seed=32)), ("asymmetric", S.random_graph_batch(16384, 32, seed=32))):
    inp = {"_atomic_numbers": b["Z"].to(dev), "_idx_i": b["idx_i"].to(dev), "_idx_j": b["idx_j"].to(dev)}
    r = b["r_ij"].to(dev).requires_grad_(True)
    def call():
        d = dict(inp); d["_Rij"] = r
        x = rep(d)["scalar_representation"]
        return torch.autograd.grad([x.sum()], [r])[0]
    for _ in range(3): call()
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_report()
    for _ in range(5): call()
    prof = _lib.profile_report(); _lib.profile_enable(False)
    print("==", name, "E =", int(b["idx_i"].shape[0]))
    tot = 0.0
    for tag, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s %5.1f launches/call %9.1f us/call" % (tag, cnt / 5, 1e3 * ms / 5)); tot += 1e3 * ms / 5
    print("   total %.1f us" % tot)
