"""Experiment (round 4): how much of the box-regime PaiNN message time is the ATOM ORDER?
Times spk_painn_message_{fwd,bwd}_f32 (row and tile families) on the 32k-atom water box with the atoms
in (a) the generator's lattice order, (b) a random order, (c) cell-sorted orders of several cell sizes, (d) Morton order."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import spk_oracle as O
from schnetpack_amd import _lib, ops, synthetic as S
dev = torch.device("cuda:0")
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 22
b = S.water_box(n_side)
N = b["Z"].shape[0]
R = b["R"].numpy(); Lbox = float(b["cell"][0, 0])
F, K = 128, 20
g = torch.Generator().manual_seed(0)
c0 = torch.randn(N, 3 * F, generator=g); q0 = torch.randn(N, F, generator=g); mu0 = torch.randn(N, 3, F, generator=g)
gq0 = torch.randn(N, F, generator=g); gmu0 = torch.randn(N, 3, F, generator=g)
wf = (torch.randn(3 * F, K, generator=g) * 0.3).to(dev); bf = (torch.randn(3 * F, generator=g) * 0.1).to(dev)
off, w = O.gaussian_rbf_params(K, 5.0); offd, wd = off.to(dev), w.to(dev)
rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, K, offd, wd, 5.0)
L = _lib.lib()

def morton(c):
    def spread(v):
        v = v.astype(np.uint64); r = np.zeros_like(v)
        for bit in range(10): r |= ((v >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit)
        return r
    return spread(c[:, 0]) << np.uint64(2) | spread(c[:, 1]) << np.uint64(1) | spread(c[:, 2])

def orders():
    yield "lattice", np.arange(N)
    if os.environ.get("EXP_ORDER_ONLY_LATTICE"): return
    yield "random", np.random.RandomState(1).permutation(N)
    for w_ in (5.0, 2.5):
        nc = int(Lbox / w_); c = np.minimum((R / (Lbox / nc)).astype(np.int64), nc - 1)
        yield "cells%.1f" % w_, np.argsort((c[:, 0] * nc + c[:, 1]) * nc + c[:, 2], kind="stable")
        yield "morton%.1f" % w_, np.argsort(morton(c), kind="stable")

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

res = {}
for name, perm in orders():            # perm[new] = old
    inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
    ii, jj = inv[b["idx_i"].numpy()], inv[b["idx_j"].numpy()]
    o = np.lexsort((jj, ii))
    ii, jj = ii[o], jj[o]
    r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])[torch.from_numpy(o)]
    pt = torch.from_numpy(perm)
    D = lambda t: t[pt].to(dev).contiguous()
    cd, qd, mud, gqd, gmud = map(D, (c0, q0, mu0, gq0, gmu0))
    rd = r.to(dev).contiguous()
    plan = ops.EdgePlan(torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev), N, rd)
    q_out = torch.empty(N, F, device=dev); mu_out = torch.empty(N, 3, F, device=dev)
    gc = torch.empty(N, 3 * F, device=dev); gmu_in = torch.empty(N, 3, F, device=dev); gr = torch.zeros(rd.shape[0], 3, device=dev)
    row = {}
    for mode, mname in ((-1, "row"), (1, "tile")):
        L.spk_painn_set_tile(mode)
        fwd = lambda: _lib.check(L.spk_painn_message_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(qd), _lib.fptr(mud), _lib.fptr(rd), _lib.fptr(wf), _lib.fptr(bf), F, _lib.fptr(q_out), _lib.fptr(mu_out), _lib.stream()))
        bwd = lambda: _lib.check(L.spk_painn_message_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(mud), _lib.fptr(gqd), _lib.fptr(gmud), _lib.fptr(rd), _lib.fptr(wf), _lib.fptr(bf), F, _lib.fptr(gc), _lib.fptr(gmu_in), _lib.fptr(gr), _lib.stream()))
        row[mname + "_fwd_us"] = round(timeit(fwd), 1); row[mname + "_bwd_us"] = round(timeit(bwd), 1)
    L.spk_painn_set_tile(0)
    # block kernels (spk_painn_blk.hip): per-tag HIP-event times of prep / forward / backward passes
    if os.environ.get("EXP_NO_BLOCKS"):
        res[name] = row; print(name, row, flush=True); continue
    okb, max_u, n_tiles = plan.build_blocks(K, F)
    row["blk_plan"] = {"ok": okb, "max_unique": max_u, "tiles": n_tiles, "sub_n_hist": torch.bincount(plan._block_bufs["sub_n"].cpu()).tolist()}
    if okb:
        L.spk_painn_set_block(1)
        row["blk_fwd_us"] = round(timeit(fwd), 1); row["blk_bwd_us"] = round(timeit(bwd), 1)
        L.spk_profile_enable(1); L.spk_profile_report()
        for _ in range(5): fwd(); bwd()
        rep = L.spk_profile_report().decode(); L.spk_profile_enable(0)
        row["blk_tags_us"] = {ln.split()[0]: round(float(ln.split()[2]) / int(ln.split()[1]) * 1e3, 1) for ln in rep.strip().splitlines()}
        # cycle stamps of a few workgroups of the forward
        dbg = torch.zeros(64, dtype=torch.int64, device=dev)
        for blk in (1001,):
            dbg.zero_(); L.spk_painn_blk_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()), blk); fwd(); torch.cuda.synchronize()
            st = dbg.cpu().tolist(); L.spk_painn_blk_set_debug_buffer(None, 0)
            print("stamps wg", blk, [(i, st[i] - st[0]) for i in sorted(range(64), key=lambda i: st[i]) if st[i]])
        L.spk_painn_set_block(0)
    res[name] = row
    print(name, row, flush=True)
print(json.dumps({"n_atoms": N, "n_edges": int(b["idx_i"].shape[0]), "orders": res}))
