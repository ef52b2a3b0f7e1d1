"""The force-matching gradient engine on the device (csrc/spk_fm.hip, torch.ops.spk_hip.schnet_fm / painn_fm): energies, forces and
every weight gradient of the force-matching loss against the float64 restatement oracle/fm_oracle.py (pinned to autograd's double
backward through the oracle of the hot path) -- what the reference obtains with ``create_graph=True`` (atomistic/response.py:59-68,
task.py:166-185).  Tolerances: energies / forces 1e-5 (north_star), weight gradients 2e-5 of each tensor's largest entry (fp32
arithmetic in a different summation order than float64; the CPU emulation of the same engine in fp32 meets the same bound)."""
import pytest
import torch

from conftest import record_value, rel_err
from oracle import fm_oracle as FM
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda", 0)


def _params(kind, F_=128, L=3, n_rbf=20, radial="gaussian", shared=False, nf=None, bias=True):
    rep_p = (O.init_schnet_params(F_, L, n_rbf, 5.0, radial=radial, n_filters=nf) if kind == "schnet"
             else O.init_painn_params(F_, L, n_rbf, 5.0, radial=radial, shared_filters=shared))
    head_p = O.init_atomwise_params(F_, seed=1)
    if bias:
        torch.manual_seed(7)
        for p in (rep_p, head_p):
            for k in list(p):
                if k.endswith("bias"):
                    p[k] = 0.1 * torch.randn_like(p[k])
    return rep_p, head_p


def _oracle(kind, rep_p, head_p, b, L, Et, Ft, shared=False, wE=0.01, wF=0.99):
    M_, N = int(b["n_mol"]), b["Z"].shape[0]
    if kind == "schnet":
        E, F, saved = FM.schnet_forward(rep_p, head_p, b, L)
    else:
        E, F, saved = FM.painn_forward(rep_p, head_p, b, L, shared_filters=shared)
    gE, gF = 2 * wE * (E - Et.double()) / M_, 2 * wF * (F - Ft.double()) / (3 * N)
    g = FM.schnet_backward(saved, gE, gF) if kind == "schnet" else FM.painn_backward(saved, gE, gF)
    loss = wE * ((E - Et.double()) ** 2).mean() + wF * ((F - Ft.double()) ** 2).mean()
    return E, F, g, float(loss)


def _device_step(kind, rep_p, head_p, b, L, Et, Ft, dev, radial="gaussian", shared=False, nf=None, n_rbf=20, F_=128, engine=True):
    from schnetpack_amd import model as M
    kw = dict(shared_filters=True) if shared else {}
    if nf is not None:
        kw["n_filters"] = nf
    model = M.build_model(kind, F_, L, n_rbf, 5.0, radial, **kw)
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).train()
    assert model.fm_engine
    model.fm_engine = engine
    out = model(M.batch_to_inputs(b, dev))
    loss = torch.ops.spk_hip.fm_loss(out["energy"], Et.to(dev), out["forces"], Ft.to(dev), 0.01, 0.99)
    loss.backward()
    grads = {k: p.grad.detach().cpu().double() for k, p in model.representation.named_parameters() if p.grad is not None}
    grads.update({k: p.grad.detach().cpu().double() for k, p in model.output_modules[0].named_parameters() if p.grad is not None})
    return out["energy"].detach().cpu().double(), out["forces"].detach().cpu().double(), grads, float(loss.detach())


def _compare(got, ref, tol_g=2e-5):
    E, F, g, loss = got
    E_o, F_o, g_o, loss_o = ref
    assert rel_err(E, E_o) < 1e-5 and rel_err(F, F_o) < 1e-5
    assert abs(loss - loss_o) / abs(loss_o) < 1e-5
    worst = ("", 0.0)
    for k, r in g_o.items():
        assert k in g, k
        e = float((g[k].reshape(r.shape) - r).abs().max()) / (float(r.abs().max()) + 1e-300)
        if e > worst[1]:
            worst = (k, e)
    record_value("fm_engine_step", "engine", "weight_gradient_worst_tensor", worst[1], tol_g)
    record_value("fm_engine_step", "engine", "forces", rel_err(F, F_o), 1e-5)
    assert worst[1] < tol_g, worst
    return worst


@pytest.mark.parametrize("kind,radial", [("schnet", "gaussian"), ("schnet", "bessel"), ("painn", "gaussian"), ("painn", "bessel")])
def test_weight_gradients_of_an_8_frame_batch_match_the_float64_oracle(dev, kind, radial):
    """configs[3] per-GPU share: 8 aspirin frames, F = 128, 3 interactions."""
    b = S.molecule_batch("aspirin", 8, seed=3)
    rep_p, head_p = _params(kind, radial=radial)
    g = torch.Generator().manual_seed(1)
    Et, Ft = torch.randn(8, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)
    ref = _oracle(kind, rep_p, head_p, b, 3, Et, Ft)
    got = _device_step(kind, rep_p, head_p, b, 3, Et, Ft, dev, radial=radial)
    _compare(got, ref)


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_lists_that_are_neither_symmetric_nor_ordered_in_j(dev, kind):
    """vesin / LAMMPS lists (transform/neighborlist.py:446-456, interfaces/lammps/pair_schnetpack.cpp:240-267) need not be symmetric and
    the neighbour index has no order: the transposed sums run over the by-neighbour CSR built on the device."""
    b = dict(S.molecule_batch("aspirin", 3, seed=4))
    E = b["idx_i"].shape[0]
    keep = torch.ones(E, dtype=torch.bool)
    keep[::5] = False
    g = torch.Generator().manual_seed(0)
    # shuffle inside every row (idx_i stays ascending)
    key = b["idx_i"].double() + 0.9 * torch.rand(E, generator=g).double()
    order = torch.argsort(key)
    for k in ("idx_i", "idx_j", "offsets"):
        b[k] = b[k][order][keep[order]]
    assert bool((b["idx_i"][1:] >= b["idx_i"][:-1]).all())
    rep_p, head_p = _params(kind, F_=64, L=2, n_rbf=12, nf=(96 if kind == "schnet" else None))
    Et, Ft = torch.randn(3, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)
    ref = _oracle(kind, rep_p, head_p, b, 2, Et, Ft)
    got = _device_step(kind, rep_p, head_p, b, 2, Et, Ft, dev, nf=(96 if kind == "schnet" else None), n_rbf=12, F_=64)
    _compare(got, ref)


def test_shared_filters_and_shared_interactions(dev):
    """painn.py:179-183 (one filter slice for all interactions) and nn/utils.py:11-18 (the same block object repeated): the engine writes one
    gradient slot per interaction, autograd sums the slots of a shared tensor."""
    from schnetpack_amd import model as M
    b = S.molecule_batch("aspirin", 2, seed=9)
    g = torch.Generator().manual_seed(2)
    Et, Ft = torch.randn(2, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)
    rep_p, head_p = _params("painn", F_=32, L=3, n_rbf=8, shared=True)
    ref = _oracle("painn", rep_p, head_p, b, 3, Et, Ft, shared=True)
    got = _device_step("painn", rep_p, head_p, b, 3, Et, Ft, dev, shared=True, n_rbf=8, F_=32)
    _compare(got, ref)
    for kind in ("schnet", "painn"):
        res = []
        for engine in (True, False):
            torch.manual_seed(0)
            model = M.build_model(kind, 32, 3, 8, 5.0, shared_interactions=True).to(dev).train()
            model.fm_engine = engine
            out = model(M.batch_to_inputs(b, dev))
            loss = torch.ops.spk_hip.fm_loss(out["energy"], Et.to(dev), out["forces"], Ft.to(dev), 0.01, 0.99)
            loss.backward()
            res.append({k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None})
        assert set(res[0]) == set(res[1])
        for k in res[0]:
            assert rel_err(res[0][k], res[1][k]) < 2e-4, (kind, k)      # (the operator-by-operator path carries its own fp32 noise)


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_engine_agrees_with_the_operator_by_operator_training_path(dev, kind):
    b = S.molecule_batch("aspirin", 4, seed=6)
    g = torch.Generator().manual_seed(3)
    Et, Ft = torch.randn(4, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)
    rep_p, head_p = _params(kind)
    a = _device_step(kind, rep_p, head_p, b, 3, Et, Ft, dev, engine=True)
    c = _device_step(kind, rep_p, head_p, b, 3, Et, Ft, dev, engine=False)
    assert rel_err(a[0], c[0]) < 1e-5 and rel_err(a[1], c[1]) < 1e-5
    for k in c[2]:
        assert rel_err(a[2][k], c[2][k]) < 2e-4, k


def test_refusals_and_flags(dev):
    """A recorded backward or a gradient w.r.t. the positions raises; a list whose idx_i is not ascending gives NaN energies (device flag)."""
    from schnetpack_amd import model as M
    b = S.molecule_batch("aspirin", 2, seed=1)
    model = M.build_model("schnet", 32, 1, 8).to(dev).train()
    out = model(M.batch_to_inputs(b, dev))
    with pytest.raises(RuntimeError, match="first-order"):
        torch.autograd.grad((out["forces"] ** 2).sum(), list(model.parameters())[:1], create_graph=True)
    bad = dict(b)
    bad["idx_i"] = torch.flip(b["idx_i"], [0])
    bad["idx_j"] = torch.flip(b["idx_j"], [0])
    bad["offsets"] = torch.flip(b["offsets"], [0])
    out = model(M.batch_to_inputs(bad, dev))
    assert bool(torch.isnan(out["energy"]).all())
    # ... and the backward SAYS so instead of handing a NaN loss to the optimizer: the message names the way out
    with pytest.raises(RuntimeError, match="not sorted ascending.*fm_engine = False"):
        (out["energy"] ** 2).sum().backward()
    # an atomic number outside the embedding table: NaN energies of its molecule (like the eval kernels), flagged in the backward
    badz = dict(b)
    badz["Z"] = b["Z"].clone()
    badz["Z"][3] = 100000
    model.zero_grad()
    out = model(M.batch_to_inputs(badz, dev))
    assert bool(torch.isnan(out["energy"][0])) and not bool(torch.isnan(out["energy"][1]))
    with pytest.raises(RuntimeError, match="outside the embedding table"):
        (out["forces"] ** 2).sum().backward()
    # the operator-by-operator path takes the unsorted list (the documented way out)
    model.fm_engine = False
    model.zero_grad()
    o1 = model(M.batch_to_inputs(bad, dev))
    o2 = model(M.batch_to_inputs(b, dev))
    assert rel_err(o1["energy"].detach().cpu(), o2["energy"].detach().cpu()) < 1e-5
    assert rel_err(o1["forces"].detach().cpu(), o2["forces"].detach().cpu()) < 1e-5


def test_transpose_plan_is_a_stable_sort_by_neighbour(dev):
    """spk_transpose_plan through the C ABI: perm = stable argsort of idx_j, colptr its CSR (column N = out-of-range neighbours)."""
    import ctypes
    from schnetpack_amd import _lib
    L = _lib.lib()
    N, E = 301, 7001
    g = torch.Generator().manual_seed(5)
    jj = torch.randint(0, N, (E,), generator=g)
    jj[17] = N + 5
    jj[4000] = -3
    d = jj.to(dev)
    L.spk_transpose_plan_bytes.restype = ctypes.c_int64
    nb = L.spk_transpose_plan_bytes(ctypes.c_int64(E), ctypes.c_int64(N))
    tmp = torch.empty(nb, dtype=torch.uint8, device=dev)
    colptr = torch.empty(N + 2, dtype=torch.int32, device=dev)
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    rc = L.spk_transpose_plan(ctypes.c_void_p(d.data_ptr()), ctypes.c_int64(E), ctypes.c_int64(N), ctypes.c_void_p(colptr.data_ptr()),
                              ctypes.c_void_p(perm.data_ptr()), ctypes.c_void_p(tmp.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    key = torch.where((jj >= 0) & (jj < N), jj, torch.full_like(jj, N))
    ref = torch.sort(key, stable=True).indices
    assert torch.equal(perm.cpu().long(), ref)
    cp = torch.searchsorted(key[ref].contiguous(), torch.arange(N + 2))
    assert torch.equal(colptr.cpu().long(), cp)


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_large_batch_gradients_are_the_mean_of_its_two_halves(dev, kind):
    """Size-independent property at a size the float64 oracle does not reach: 320 aspirin frames (6 720 atoms, 860 k (atom, channel) items --
    the slotted row kernels walk their grid-stride loop twice, the Dense layers of the pair rows take the many-tile kernels, every
    weight-gradient problem is cut into row slices) against the SAME weights on the two halves of the batch.  Both MSE terms are means over
    molecules / atoms, so loss_full = (loss_a + loss_b) / 2 and every gradient likewise (task.py:142-146)."""
    n = 320
    full = S.molecule_batch("aspirin", n, seed=12)
    na = full["Z"].shape[0] // 2
    g = torch.Generator().manual_seed(4)
    Et, Ft = torch.randn(n, generator=g), torch.randn(full["Z"].shape[0], 3, generator=g)

    def half(lo):
        atoms = slice(lo * na, (lo + 1) * na)
        keep = (full["idx_i"] >= lo * na) & (full["idx_i"] < (lo + 1) * na)
        b = {"Z": full["Z"][atoms], "R": full["R"][atoms], "idx_m": full["idx_m"][atoms] - lo * (n // 2),
             "idx_i": full["idx_i"][keep] - lo * na, "idx_j": full["idx_j"][keep] - lo * na, "offsets": full["offsets"][keep], "n_mol": n // 2}
        return b, Et[lo * (n // 2):(lo + 1) * (n // 2)], Ft[atoms]

    rep_p, head_p = _params(kind)
    E, F, gr, loss = _device_step(kind, rep_p, head_p, full, 3, Et, Ft, dev)
    parts = []
    for lo in (0, 1):
        b, et, ft = half(lo)
        parts.append(_device_step(kind, rep_p, head_p, b, 3, et, ft, dev))
    assert rel_err(E, torch.cat([p[0] for p in parts])) < 1e-6 and rel_err(F, torch.cat([p[1] for p in parts])) < 1e-6
    assert abs(loss - 0.5 * (parts[0][3] + parts[1][3])) / abs(loss) < 1e-5
    worst = ("", 0.0)
    for k, v in gr.items():
        r = 0.5 * (parts[0][2][k] + parts[1][2][k])
        e = float((v - r).abs().max()) / (float(r.abs().max()) + 1e-300)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 5e-5, worst


@pytest.mark.parametrize("kind,frames", [("schnet", 8), ("painn", 8), ("painn", 3), ("schnet", 5)])
def test_row_chains_equal_the_launch_by_launch_step(dev, kind, frames):
    """Round 5 EXPERIMENT (opt-in, spk_fm_set_chain(1)): the atom-local launches of a pass recorded as row chains (csrc/spk_fm_chain.h: one
    workgroup per 4 atoms walks Dense / element-wise stages on v_mfma_f32_4x4x1) against the same step launch by launch: energies, forces and
    every weight gradient agree to fp32 rounding, the chained step takes far fewer launches (and, measured, more time:
    profiles/r05_row_chains.md -- which is why it is off by default).  The chained step is also held against the float64 oracle.
    frames = 3 / 5: the last workgroup owns fewer than four atoms (63 = 15 x 4 + 3; 105 = 26 x 4 + 1)."""
    from schnetpack_amd import _lib
    b = S.molecule_batch("aspirin", frames, seed=21)
    g = torch.Generator().manual_seed(5)
    Et, Ft = torch.randn(frames, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)
    rep_p, head_p = _params(kind)
    res, launches = {}, {}
    try:
        for mode in (0, 1):
            _lib.lib().spk_fm_set_chain(mode)
            _lib.profile_enable(True); _lib.profile_report()
            res[mode] = _device_step(kind, rep_p, head_p, b, 3, Et, Ft, dev)
            prof = _lib.profile_report(); _lib.profile_enable(False)
            launches[mode] = sum(c for c, _ in prof.values())
            if mode == 1:
                assert "fm_chain" in prof
    finally:
        _lib.lib().spk_fm_set_chain(-1)
    _compare(res[1], _oracle(kind, rep_p, head_p, b, 3, Et, Ft))
    a, c = res[0], res[1]
    assert rel_err(c[0], a[0]) < 2e-6 and rel_err(c[1], a[1]) < 2e-6 and abs(c[3] - a[3]) / abs(a[3]) < 2e-6
    for k in a[2]:
        assert float((c[2][k] - a[2][k]).abs().max()) / (float(a[2][k].abs().max()) + 1e-300) < 5e-6, k
    assert launches[1] < 0.75 * launches[0], launches
