"""Debug aid (round 6): saved tensors of the SchNet forward (h | pre3 per interaction, then the raw filter outputs g) with the split
path on / off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
b = S.molecule_batch("aspirin", 7, seed=3)
L = 3
rep = O.init_schnet_params(128, L, 20, 5.0, radial="gaussian")
head = O.init_atomwise_params(128, seed=1)
m = M.build_model("schnet", 128, L, 20, 5.0, "gaussian")
M.load_reference_params(m, rep, head)
m = m.to(dev).eval()
r = m.representation
inp = M.batch_to_inputs(b, dev)
R = inp["_positions"]
r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]]).contiguous()
x0 = r.embedding(inp["_atomic_numbers"])
ws = r.interaction_weights()
kind, p0, p1 = r.radial_basis.kernel_params()
res = {}
for split in (0, 1):
    _lib.set_split(split)
    x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
    torch.cuda.synchronize()
    res[split] = (x.cpu(), saved.cpu())
N = x0.shape[0]
s0, s1 = res[0][1], res[1][1]
print("saved floats", s0.numel(), "N", N)
for l in range(L):
    h0 = s0[l * N * 256: l * N * 256 + N * 128].view(N, 128); h1 = s1[l * N * 256: l * N * 256 + N * 128].view(N, 128)
    p0_ = s0[l * N * 256 + N * 128: (l + 1) * N * 256].view(N, 128); p1_ = s1[l * N * 256 + N * 128: (l + 1) * N * 256].view(N, 128)
    eh = (h0 - h1).abs(); ep = (p0_ - p1_).abs()
    print("layer", l, "h diff max %.3e" % eh.max(), " pre3 diff max %.3e (max |pre3| %.2f)" % (ep.max(), p0_.abs().max()), " atoms with pre3 diff > 3e-6:", sorted(set(torch.nonzero(ep > 3e-6)[:, 0].tolist())))
g0 = s0[L * N * 256:]; g1 = s1[L * N * 256:]
gsz = g0.numel() // L
for l in range(L):
    a = g0[l * gsz:(l + 1) * gsz].view(-1, 128); c = g1[l * gsz:(l + 1) * gsz].view(-1, 128)
    e = (a - c).abs()
    rows = sorted(set(torch.nonzero(e > 3e-6)[:, 0].tolist()))
    print("layer", l, "g diff max %.3e (max |g| %.2f)" % (e.max(), a.abs().max()), "rows > 3e-6:", rows[:20])
    for rr in rows[:4]:
        print("    row", rr, "g0", a[rr, :6].tolist(), "g1", c[rr, :6].tolist())
_lib.set_split(1)
