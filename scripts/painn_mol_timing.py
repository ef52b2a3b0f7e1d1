"""Phase timing of the molecule-resident PaiNN kernels (cycle stamps of thread 0 / workgroup 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from schnetpack_amd import _lib, model as M, synthetic as S
dev = torch.device("cuda:0")
b = S.molecule_batch("aspirin", int(sys.argv[1]) if len(sys.argv) > 1 else 256, seed=0)
torch.manual_seed(0)
m = M.build_model("painn").to(dev).eval()
inp = M.batch_to_inputs(b, dev)
L = _lib.lib()
dbg = torch.zeros(512, dtype=torch.int64, device=dev)
for rep in range(3):
    dbg.zero_()
    L.spk_painn_mol_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    out = m(dict(inp))
    torch.cuda.synchronize()
L.spk_painn_mol_set_debug_buffer(None)
st = dbg.cpu().tolist()
names = {0: "fwd group start"}
for l in range(3):
    for k, n in enumerate(["set-up / prev done", "P1 ctx.0 done", "P2 ctx.1 (c) done", "P3 message rows done", "P3 written", "P4 mix done", "P5 ictx.0 done", "P6 update done"]):
        names[1 + 8 * l + k] = "L%d %s" % (l, n)
for base in (0, 64):
    prev = st[base]
    for k in sorted(names):
        kk = k + base
        if kk < len(st) and st[kk]:
            print("  %-36s %8d  (+%d)" % (names[k] if base == 0 else "bwd " + str(k), st[kk] - st[base], st[kk] - prev)); prev = st[kk]
print("bwd M2 detail (top layer): phase start -> MFMAs retired %d, -> partial barrier %d, -> epilogue done %d" % (st[64 + 40] - st[64 + 2], st[64 + 41] - st[64 + 2], st[64 + 42] - st[64 + 2]))
print("   per wave: weights arrived (rel. to M1 start)", [st[64 + 52 + w] - st[64 + 1] for w in range(8)], " MFMAs retired (rel. to M2 start)", [st[64 + 44 + w] - st[64 + 2] for w in range(8)])
print("bwd message (top layer), per wave, cycles from the wave's entry: [atom start, edge loop done, results stored] x atoms, exit")
for w in range(8):
    e = st[128 + 16 * w]
    print("   wave %d (entry %+6d vs wave 0):" % (w, e - st[128]), [st[128 + 16 * w + k] - e if st[128 + 16 * w + k] else None for k in range(1, 14)])
print("HW_ID per wave (simd = bits 4-5):", [(hex(x), (x >> 4) & 3) for x in st[200:208]])
_lib.profile_enable(True); _lib.profile_report()
for _ in range(20):
    m(dict(inp))
print({k: round(1e3 * v[1] / v[0], 1) for k, v in _lib.profile_report().items()})
