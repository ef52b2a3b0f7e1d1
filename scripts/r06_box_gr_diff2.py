"""Debug aid: as r06_box_gr_diff.py for L = 1, 2, 3 interactions; also split-vs-split determinism."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
b = S.water_box(n_side=10, seed=3)
ii, jj = b["idx_i"], b["idx_j"]
for L in (1, 2, 3):
    rep = O.init_schnet_params(128, L, 20, 5.0); head = O.init_atomwise_params(128, seed=1)
    m = M.build_model("schnet", 128, L, 20, 5.0); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
    r = m.representation
    inp = M.batch_to_inputs(b, dev)
    R = inp["_positions"]
    r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]] + inp["_offsets"]).contiguous()
    x0 = r.embedding(inp["_atomic_numbers"]).detach()
    ws = r.interaction_weights(); kind, p0, p1 = r.radial_basis.kernel_params()
    gx = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    outs = []
    for sp in (0, 1, 1, 1):
        _lib.set_split(sp)
        x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
        gr, gx0 = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, L == 2)
        torch.cuda.synchronize()
        outs.append(gr.cpu().clone())
    scale = outs[0].abs().max()
    for k in (1, 2, 3):
        dg = (outs[0] - outs[k]).abs().max(1).values
        bad = torch.nonzero(dg > 1e-4 * scale).flatten()
        print("L=%d want_gx0=%s run %d: %d bad edges (i<j: %d), max %.2e" % (L, L == 2, k, bad.numel(), int((ii[bad] < jj[bad]).sum()), float(dg.max() / scale)))
    dd = (outs[1] - outs[2]).abs().max(1).values; d3 = (outs[1] - outs[3]).abs().max(1).values
    print("   split-vs-split: edges differing > 1e-4: %d / %d" % (int((dd > 1e-4 * scale).sum()), int((d3 > 1e-4 * scale).sum())))
_lib.set_split(1)
