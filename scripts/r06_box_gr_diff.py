"""Debug aid (round 6): dL/dr_ij of the general SchNet driver on a water box with the split path on / off: which pairs differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
rep = O.init_schnet_params(); head = O.init_atomwise_params(128, seed=1)
m = M.build_model("schnet"); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
r = m.representation
b = S.water_box(n_side=10, seed=3)
inp = M.batch_to_inputs(b, dev)
R = inp["_positions"]
r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]] + inp["_offsets"]).contiguous()
x0 = r.embedding(inp["_atomic_numbers"]).detach()
ws = r.interaction_weights(); kind, p0, p1 = r.radial_basis.kernel_params()
gx = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
res = {}
for sp in (0, 1):
    _lib.set_split(sp)
    x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
    gr, gx0 = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, False)
    torch.cuda.synchronize()
    res[sp] = (x.cpu(), gr.cpu())
dx = (res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()
dg = (res[0][1] - res[1][1]).abs().max(1).values
scale = res[0][1].abs().max()
bad = torch.nonzero(dg > 1e-4 * scale).flatten()
print("x diff %.2e; gr: %d of %d edges off by > 1e-4 (max %.2e)" % (float(dx), bad.numel(), dg.numel(), float(dg.max() / scale)))
# position of the bad edges in the list of canonical pairs (i < j by edge order): tile = position // 32
ii, jj = b["idx_i"], b["idx_j"]
E = ii.shape[0]
print("bad edges (first 20):", bad[:20].tolist())
print("their (i, j):", [(int(ii[e]), int(jj[e])) for e in bad[:10]])
frac = bad.double() / E
print("relative position in the edge list: min %.3f max %.3f" % (float(frac.min()), float(frac.max())) if bad.numel() else "")
import collections
print("histogram of bad edges by tenth of the list:", collections.Counter((frac * 10).long().tolist()))
_lib.set_split(1)
g0, g1 = res[0][1], res[1][1]
for e in bad[:12].tolist():
    n0, n1 = float(g0[e].norm()), float(g1[e].norm())
    print("edge %6d (i %4d j %4d) |gr| fp32 %.4e split %.4e ratio %.4f  d %.4f" % (e, int(ii[e]), int(jj[e]), n0, n1, n1 / n0, float(r_ij[e].norm())))
# which canonical-pair tiles?  (positions in the half list: edges with i < j in list order)
lt = int((ii[bad] < jj[bad]).sum())
print("bad edges with i < j: %d, i > j: %d" % (lt, bad.numel() - lt))
print("distinct centre atoms i of bad edges:", sorted(set(ii[bad].tolist()))[:40])
print("distinct neighbours j of bad edges:", sorted(set(jj[bad].tolist()))[:40])
if os.environ.get("VISITS"):
    import ctypes
    nt = (E // 2 + 31) // 32
    dbg = torch.zeros(nt + 64, dtype=torch.int64, device=dev)
    _lib.lib().spk_cfconv_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    _lib.set_split(1)
    x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
    gr, gx0 = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, False)
    torch.cuda.synchronize()
    _lib.lib().spk_cfconv_set_debug_buffer(None)
    v = dbg.cpu()[:nt]
    print("tiles", nt, "visit counts:", {int(k): int((v == k).sum()) for k in v.unique()})
