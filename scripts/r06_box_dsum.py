"""Debug aid (round 6): dump (s1, s2, e, e2) of every pair from the split geometry backward (build with -DSPK_DBG_DSUM), one interaction,
and compare with what lands in gr and with the fp32 path: is a bad edge a wrong sum or a wrong store?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
b = S.water_box(n_side=10, seed=3)
L = int(os.environ.get("LAYERS", "1"))
rep = O.init_schnet_params(128, L, 20, 5.0); head = O.init_atomwise_params(128, seed=1)
m = M.build_model("schnet", 128, L, 20, 5.0); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
r = m.representation
inp = M.batch_to_inputs(b, dev)
R = inp["_positions"]
r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]] + inp["_offsets"]).contiguous()
x0 = r.embedding(inp["_atomic_numbers"]).detach()
ws = r.interaction_weights(); kind, p0, p1 = r.radial_basis.kernel_params()
gx = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
E = r_ij.shape[0]
nt = (E // 2 + 31) // 32
def run(sp, dump):
    _lib.set_split(sp)
    dbg = torch.zeros(nt * 32 * 4 + nt * 64 * 4 + 64, dtype=torch.float32, device=dev)
    x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
    if dump:
        _lib.lib().spk_cfconv_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    gr, gx0 = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, L == 2)
    torch.cuda.synchronize()
    _lib.lib().spk_cfconv_set_debug_buffer(None)
    return gr.detach().cpu(), dbg.cpu()[: nt * 32 * 4].view(-1, 4), dbg.cpu()[nt * 32 * 4: nt * 32 * 4 + nt * 64 * 4].view(nt, 64, 4)
g0, _, _ = run(0, False)
parts = []; e_all = []
scale = float(g0.abs().max())
rc = r_ij.detach().cpu()
for it in range(int(os.environ.get("RUNS", "5"))):
    g1, d, pp = run(1, True)
    parts.append(pp); e_all.append(d[:, 2].view(torch.int32).long())
    e = d[:, 2].view(torch.int32).long(); e2 = d[:, 3].view(torch.int32).long()
    ok = (d[:, 0] != 0) | (d[:, 1] != 0)
    e, e2, s1, s2 = e[ok], e2[ok], d[ok, 0], d[ok, 1]
    want1 = s1[:, None] * rc[e]; want2 = -s2[:, None] * rc[e]
    st1 = (g1[e] - want1).abs().max(1).values; st2 = (g1[e2] - want2).abs().max(1).values
    c1 = (g0[e] - want1).abs().max(1).values; c2 = (g0[e2] - want2).abs().max(1).values
    tot = (g0 - g1).abs().max(1).values
    print("run %d: dumped pairs %d of %d; gr != dumped sums (store): e %d, e2 %d; dumped sums != fp32 path (compute): e %d, e2 %d; gr != fp32: %d" % (
        it, int(ok.sum()), E // 2, int((st1 > 1e-4 * scale).sum()), int((st2 > 1e-4 * scale).sum()),
        int((c1 > 1e-4 * scale).sum()), int((c2 > 1e-4 * scale).sum()), int((tot > 1e-4 * scale).sum())))
    badp = torch.nonzero(c2 > 1e-4 * scale).flatten()
    for p in badp[:8].tolist():
        print("   pair slot %d (tile %d lane %d) e %d e2 %d  s2 dumped %.5e  fp32 %.5e" % (p, p // 32, p % 32, int(e[p]), int(e2[p]), float(s2[p]),
              float(-(g0[e2[p]] * rc[e[p]]).sum() / (rc[e[p]] ** 2).sum())))
_lib.set_split(1)
# per-lane, per-t running sums of dsum2: which lanes / which t differ between runs?  (median of the runs = reference)
ii, jj = b["idx_i"], b["idx_j"]
ref = torch.stack(parts).median(0).values
for it, pp in enumerate(parts):
    df = (pp - ref).abs() > 1e-4 * ref.abs().max()
    tiles = torch.nonzero(df.any(2).any(1)).flatten().tolist()
    for tl in tiles[:6]:
        lanes = torch.nonzero(df[tl].any(1)).flatten().tolist()
        first_t = [int(torch.nonzero(df[tl, l]).flatten()[0]) for l in lanes]
        print("run %d tile %d: lanes off %s; first running sum off (t) %s" % (it, tl, lanes, first_t))
        es = [int(e_all[it][tl * 32 + (l & 31)]) for l in lanes]
        print("      centre atoms i of those lanes:", [int(ii[e]) for e in es], " j:", [int(jj[e]) for e in es])
