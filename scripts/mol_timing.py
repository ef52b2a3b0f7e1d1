"""Phase timing of the molecule-resident SchNet kernels (cycle stamps of thread 0 / workgroup 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from schnetpack_amd import _lib, model as M, synthetic as S
dev = torch.device("cuda:0")
b = S.molecule_batch("aspirin", int(sys.argv[1]) if len(sys.argv) > 1 else 256, seed=0)
torch.manual_seed(0)
m = M.build_model("schnet").to(dev).eval()
inp = M.batch_to_inputs(b, dev)
L = _lib.lib()
dbg = torch.zeros(128 + 4 * 1024, dtype=torch.int64, device=dev)
for rep in range(3):
    dbg.zero_()
    L.spk_schnet_mol_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    out = m(dict(inp))
    torch.cuda.synchronize()
L.spk_schnet_mol_set_debug_buffer(None)
st = dbg.cpu().tolist()
names = {0: "fwd start", 31: "fwd end", 32: "bwd start", 63: "bwd end"}
for l in range(3):
    names.update({1 + 5 * l: "L%d staged" % l, 2 + 5 * l: "L%d pair tiles done" % l, 4 + 5 * l: "L%d partial sums in LDS" % l, 5 + 5 * l: "L%d C1 done" % l})
    names.update({33 + 6 * l: "bwd L%d D1 done" % (2 - l), 34 + 6 * l: "bwd L%d D2 done" % (2 - l), 35 + 6 * l: "bwd L%d E (thread 0 out)" % (2 - l),
                  36 + 6 * l: "bwd L%d E barrier" % (2 - l), 37 + 6 * l: "bwd L%d G done" % (2 - l)})
for base in (0, 32):
    prev = st[base]
    for k in sorted(names):
        if k >= base and k < base + 32 and st[k]:
            print("  %-36s %8d  (+%d)" % (names[k], st[k] - st[base], st[k] - prev)); prev = st[k]
import numpy as np
print("bwd set-up of block 0 (cycles from the first instruction):", {k: st[k] - st[130] for k in (120, 121, 122, 123, 124, 32) if st[k]})
blk = np.array(st[128:128 + 4 * 256]).reshape(256, 4)
t0 = blk[:, 0].min()
print("bwd blocks (100 MHz real time, us): start min/max %.2f %.2f   end min/median/max %.2f %.2f %.2f" % (
    (blk[:, 0].min() - t0) / 100, (blk[:, 0].max() - t0) / 100, (blk[:, 1].min() - t0) / 100, np.median(blk[:, 1] - t0) / 100, (blk[:, 1].max() - t0) / 100))
cyc = blk[:, 3] - blk[:, 2]
dur = (blk[:, 1] - blk[:, 0]) / 100.0
print("bwd per-block cycles min/median/max %d %d %d; duration us min/median/max %.1f %.1f %.1f; clock GHz median %.3f" % (
    cyc.min(), np.median(cyc), cyc.max(), dur.min(), np.median(dur), dur.max(), np.median(cyc / dur) / 1e3))
slow = np.argsort(-blk[:, 1])[:8]
print("latest blocks:", [(int(b), round((blk[b, 0] - t0) / 100, 1), round((blk[b, 1] - t0) / 100, 1), int(cyc[b])) for b in slow])
_lib.profile_enable(True); _lib.profile_report()
for _ in range(20):
    m(dict(inp))
print({k: round(1e3 * v[1] / v[0], 1) for k, v in _lib.profile_report().items()})
