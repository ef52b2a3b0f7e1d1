"""Phase timing of the pair cfconv kernels (cycle stamps of wave 0 / workgroup 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import spk_oracle as O
from schnetpack_amd import _lib, ops, synthetic as S
dev = torch.device("cuda:0")
b = S.molecule_batch("aspirin", 256, seed=0)
r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"]).to(dev)
N = b["Z"].shape[0]
plan = ops.EdgePlan(b["idx_i"].to(dev), b["idx_j"].to(dev), N, r)
off, w = O.gaussian_rbf_params(20, 5.0); offd, wd = off.to(dev), w.to(dev)
rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, 20, offd, wd, 5.0)
g = torch.Generator().manual_seed(0)
nf = 128
h = torch.randn(N, nf, generator=g).to(dev); gy = torch.randn(N, nf, generator=g).to(dev)
w1 = (torch.randn(nf, 20, generator=g) * .3).to(dev); b1 = torch.zeros(nf, device=dev)
w2 = (torch.randn(nf, nf, generator=g) / 11).to(dev); b2 = torch.zeros(nf, device=dev)
y = torch.empty(N, nf, device=dev); gh = torch.empty(N, nf, device=dev); gr = torch.zeros(r.shape[0], 3, device=dev)
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
L = _lib.lib()
names = {0: "start", 1: "staged", 2: "rbf done", 3: "gemm1+act done", 20: "tile end", 21: "kernel end"}
for t in range(4):
    names[4 + 3 * t] = "t%d gemm2 done" % t; names[5 + 3 * t] = "t%d modulate+lds" % t; names[6 + 3 * t] = "t%d reduce done" % t
for which in ("fwd", "bwd"):
    for rep in range(3):
        dbg.zero_()
        L.spk_cfconv_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
        if which == "fwd":
            _lib.check(L.spk_schnet_cfconv_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(h), _lib.fptr(r), _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), nf, _lib.fptr(y), _lib.stream()))
        else:
            _lib.check(L.spk_schnet_cfconv_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(h), _lib.fptr(gy), _lib.fptr(r), _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), nf, _lib.fptr(gh), _lib.fptr(gr), _lib.stream()))
        torch.cuda.synchronize()
    L.spk_cfconv_set_debug_buffer(None)
    st = dbg.cpu().tolist()
    print("==", which, "(cycles since kernel start; s_memtime @100MHz-ish or shader clock)")
    prev = st[0]
    for k in sorted(names):
        if st[k]:
            print("  %-18s %8d  (+%d)" % (names[k], st[k] - st[0], st[k] - prev)); prev = st[k]
