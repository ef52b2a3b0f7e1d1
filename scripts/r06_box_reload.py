import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S, _lib, model as M
dev = torch.device("cuda")
b = S.water_box(n_side=10, seed=3)
rep = O.init_schnet_params(); head = O.init_atomwise_params(128, seed=1)
m = M.build_model("schnet"); M.load_reference_params(m, rep, head); m = m.to(dev).eval()
r = m.representation
inp = M.batch_to_inputs(b, dev)
R = inp["_positions"]
r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]] + inp["_offsets"]).contiguous()
x0 = r.embedding(inp["_atomic_numbers"]).detach()
ws = r.interaction_weights(); kind, p0, p1 = r.radial_basis.kernel_params()
gx = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
_lib.set_split(0)
x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
gr0, _ = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, False)
gr0 = gr0.detach().clone(); scale = float(gr0.abs().max())
_lib.set_split(1)
for it in range(30):
    dbg.zero_()
    _lib.lib().spk_cfconv_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
    gr, gx0 = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, False)
    torch.cuda.synchronize()
    bad = int(((gr.detach() - gr0).abs().max(1).values > 1e-4 * scale).sum())
    print("mismatching reloaded values: h[i] %d  gy[j] %d  h[j] %d  gy[i] %d;  bad edges in gr: %d" % (tuple(dbg.cpu()[:4].tolist()) + (bad,)))
_lib.lib().spk_cfconv_set_debug_buffer(None)
