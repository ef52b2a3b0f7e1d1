"""Phase timing of the fused Dense-chain kernel (cycle stamps of thread 0 / workgroup 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from schnetpack_amd import _lib
from test_gpu_ops import _chain_struct, _pack
dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator().manual_seed(0)
m, F = int(sys.argv[1]) if len(sys.argv) > 1 else 5376, 128
D = lambda t: t.to(dev).contiguous()
y, x = D(torch.randn(m, F, generator=g)), D(torch.randn(m, F, generator=g))
ws = [_pack(D(torch.randn(F, F, generator=g) / 11), 0) for _ in range(3)]
bs = [D(torch.randn(F, generator=g) * 0.1) for _ in range(2)]
pre, xo, ho = (torch.empty(m, F, device=dev) for _ in range(3))
layers = [dict(w=ws[0], b=bs[0], pre_out=pre, k=F, n_out=F, act=_lib.SPK_ACT_SSP, trans=2),
          dict(w=ws[1], b=bs[1], res=x, out=xo, k=F, n_out=F, act=0, trans=2),
          dict(w=ws[2], out=ho, k=F, n_out=F, act=0, trans=2)]
c = _chain_struct(_lib, m, y, layers)
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
names = {0: "start", 1: "input staged", 12: "kernel end"}
for l in range(3):
    names[2 + 3 * l] = "L%d mfma done" % l; names[3 + 3 * l] = "L%d epilogue done" % l; names[4 + 3 * l] = "L%d barrier" % l
for rows in (16, 32):
    L.spk_chain_set_rows(rows)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        dbg.zero_()
        L.spk_chain_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
        _lib.check(L.spk_dense_chain_f32(ctypes.byref(c), _lib.stream()))
        torch.cuda.synchronize()
    L.spk_chain_set_debug_buffer(None)
    ev0.record()
    for _ in range(20):
        _lib.check(L.spk_dense_chain_f32(ctypes.byref(c), _lib.stream()))
    ev1.record(); torch.cuda.synchronize()
    st = dbg.cpu().tolist()
    print("== rows", rows, "m", m, " back-to-back launch: %.1f us" % (1e3 * ev0.elapsed_time(ev1) / 20))
    prev = st[0]
    for k in sorted(names):
        if st[k]:
            print("  %-18s %8d  (+%d)" % (names[k], st[k] - st[0], st[k] - prev)); prev = st[k]
L.spk_chain_set_rows(0)
