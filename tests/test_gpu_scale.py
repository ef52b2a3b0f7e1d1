"""Force-call parity at the scale of configs[4] (a periodic bulk-water box, >= 2^19 directed edges), under the
AUTOMATIC kernel dispatch -- the only size at which the tile-kernel dispatch (``n_edges >= 2^19``,
representation/painn.py), the skin / no-skin decision, the 10^8 float atomics of a cfconv launch and the 32-bit
offset guards are live.  Checker: the CPU oracle in float64 on the host (the reference's algorithm,
representation/painn.py:207-256, representation/schnet.py:147-173), on the very boxes ``bench.py --workload water``
times.  Tolerance 1e-5 relative (north_star), relative = max|a-b| / max|b|.

* mid box   (15^3 molecules = 10 125 atoms, ~5.4e5 edges): both models, exact list and a 1 A skin list;
* full box  (22^3 molecules = 31 944 atoms, ~1.71e6 edges, the bench workload itself): both models, once;
* supercell (size-independent property): the 11^3 box replicated 2x2x2 is the same crystal -- per-atom forces of the
  31 944-atom supercell equal the tiled forces of the 3 993-atom cell, the energy is 8x.
"""
import os

import pytest
import torch

from conftest import rel_err
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def host_threads():
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    yield
    torch.set_num_threads(old)


def _host_mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _params(kind):
    rep = O.init_schnet_params() if kind == "schnet" else O.init_painn_params()
    return rep, O.init_atomwise_params(128, seed=1)


def _model(kind, dev, rep_p, head_p):
    from schnetpack_amd import _lib, model as M
    assert _lib.get_variant() == _lib.VARIANT_AUTO
    m = M.build_model(kind)
    M.load_reference_params(m, rep_p, head_p)
    return m.to(dev).eval()


def _oracle(kind, rep_p, head_p, batch):
    # float64 needs ~75 KB per directed edge for PaiNN (autograd keeps the [E, 3F n_int] filter tensor and the per-layer
    # message tensors); fall back to the reference's own fp32 arithmetic on a small host
    E = int(batch["idx_i"].shape[0])
    dtype = torch.float64 if _host_mem_gb() > 1.5 * 75e3 * E / 1e9 + 8 else torch.float32
    return O.energy_and_forces(kind, rep_p, head_p, batch, 3, dtype=dtype)


def _gpu_call(model, batch, dev):
    from schnetpack_amd import model as M
    out = model(M.batch_to_inputs(batch, dev))
    return out["energy"].detach().cpu(), out["forces"].detach().cpu()


_MID = {}


def _mid_box():
    if not _MID:
        b = S.water_box(n_side=15, seed=3)
        assert b["idx_i"].shape[0] >= (1 << 19), b["idx_i"].shape
        _MID["box"] = b
    return _MID["box"]


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_mid_box_force_call_auto_dispatch(dev, kind):
    from schnetpack_amd import neighborlist as NL, ops
    b = _mid_box()
    rep_p, head_p = _params(kind)
    ref = _oracle(kind, rep_p, head_p, b)
    model = _model(kind, dev, rep_p, head_p)
    e, f = _gpu_call(model, b, dev)
    assert rel_err(e, ref["energy"]) < TOL
    assert rel_err(f, ref["forces"]) < TOL
    # the same box through the device neighbour list with a 1 A skin (MD lists): 1.7x the pairs, the ones beyond the
    # cutoff contribute exactly zero, so the oracle result of the exact list is the truth for it as well
    cell = b["cell"].reshape(1, 3, 3).to(dev)
    nl = NL.neighbor_list(b["R"].to(dev), 6.0, idx_m=b["idx_m"].to(dev), cell=cell,
                          pbc=torch.tensor([True, True, True], device=dev), n_systems=1)
    skin = dict(b)
    skin["idx_i"], skin["idx_j"], skin["offsets"] = nl["_idx_i"].cpu(), nl["_idx_j"].cpu(), nl["_offsets"].cpu()
    assert skin["idx_i"].shape[0] > 1.4 * b["idx_i"].shape[0]
    e2, f2 = _gpu_call(model, skin, dev)
    assert rel_err(e2, ref["energy"]) < TOL
    assert rel_err(f2, ref["forces"]) < TOL
    if kind == "painn":
        # round 6: between positions and forces the box-regime PaiNN force call has no float atomic left (row-tile message kernels, row passes
        # for the pair vectors): the forces of repeated calls are bit-identical, on the exact list and on the skin list.  (The ONE energy of a
        # 10 125-atom system is still summed with atomics by the head: last-bit differences, scripts/r06_box_repro.py.)
        for batch, (e_ref, f_ref) in ((b, (e, f)), (skin, (e2, f2))):
            for _ in range(3):
                e3, f3 = _gpu_call(model, batch, dev)
                assert torch.equal(f3, f_ref)
                assert rel_err(e3, e_ref) < 1e-6


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_full_box_force_call_auto_dispatch(dev, kind):
    """The bench workload itself (``bench.py --workload water``: 31 944 atoms, 1.71 M directed edges)."""
    b = S.water_box(n_side=22, seed=0)
    assert b["Z"].shape[0] == 31944
    rep_p, head_p = _params(kind)
    model = _model(kind, dev, rep_p, head_p)
    e, f = _gpu_call(model, b, dev)
    ref = _oracle(kind, rep_p, head_p, b)
    assert rel_err(e, ref["energy"]) < TOL
    assert rel_err(f, ref["forces"]) < TOL
    # graph-captured replay of the same call (what the bench times) reproduces it
    from schnetpack_amd.forcecall import GraphedForceCall
    from schnetpack_amd import model as M
    gc = GraphedForceCall(model)
    inp = M.batch_to_inputs(b, dev)
    for _ in range(3):
        out = gc(inp)
    torch.cuda.synchronize()
    assert rel_err(out["forces"].cpu(), ref["forces"]) < TOL


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_supercell_forces_tile(dev, kind):
    from schnetpack_amd import neighborlist as NL
    base = S.water_box(n_side=11, seed=7)
    # coordinates and box edge on a 2^-16 A grid: the shifted copies R + n L are then EXACT in float32 (< 128 A), so the
    # supercell is bit-for-bit the same crystal and any deviation is the kernels', not input rounding
    q = 65536.0
    L = round(float(base["cell"][0, 0]) * q) / q
    Rs = torch.remainder(torch.round(base["R"].double() * q) / q, L).float()
    pbc = torch.tensor([True, True, True], device=dev)

    def box(R, edge, Z):
        idx_m = torch.zeros(R.shape[0], dtype=torch.long)
        cell = (torch.eye(3) * edge).reshape(1, 3, 3)
        nl = NL.neighbor_list(R.to(dev), 5.0, idx_m=idx_m.to(dev), cell=cell.to(dev), pbc=pbc, n_systems=1)
        return {"Z": Z, "R": R, "idx_i": nl["_idx_i"].cpu(), "idx_j": nl["_idx_j"].cpu(), "offsets": nl["_offsets"].cpu(),
                "idx_m": idx_m, "n_mol": 1}
    small = box(Rs, L, base["Z"])
    shifts = torch.tensor([[a, b_, c] for a in (0, 1) for b_ in (0, 1) for c in (0, 1)], dtype=torch.float32) * L
    Rb = (Rs[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    assert Rb.shape[0] == 31944 and torch.equal((Rb.double() - shifts.double().repeat_interleave(Rs.shape[0], 0)).float(), Rs.repeat(8, 1))
    big = box(Rb, 2 * L, base["Z"].repeat(8))
    assert int(big["idx_i"].shape[0]) == 8 * int(small["idx_i"].shape[0]) >= (1 << 19)
    rep_p, head_p = _params(kind)
    ref = _oracle(kind, rep_p, head_p, small)
    model = _model(kind, dev, rep_p, head_p)
    e, f = _gpu_call(model, big, dev)
    assert rel_err(f, ref["forces"].repeat(8, 1).float()) < TOL
    assert abs(float(e[0]) - 8.0 * float(ref["energy"][0])) <= 2e-5 * abs(8.0 * float(ref["energy"][0]))


def test_tabulated_filter_experiment_force_call_on_the_water_box(dev):
    """EXPERIMENT, default off (schnetpack_amd/tabulate.py): SchNet with the filters read from 512-knot cubic-Hermite tables on the
    10 125-atom periodic box -- forward AND first-order backward through the table kernels (profile tags), energies and forces
    against the float64 oracle.  The value error of the table is ~4e-8, its slope error ~1.6e-6 of max |dW/dd| (512 knots; knot differences formed in float64): the
    forces stay inside the 1e-5 bar with the margin of the fp32-MFMA contract path, which this test runs beside it."""
    from schnetpack_amd import _lib, model as M, tabulate
    rep_p, head_p = _params("schnet")
    model = _model("schnet", dev, rep_p, head_p)
    b = S.water_box(n_side=15, seed=1)
    ref = O.energy_and_forces("schnet", rep_p, head_p, b, 3, dtype=torch.float64)
    inp = M.batch_to_inputs(b, dev)
    out0 = model(dict(inp))
    e0, f0 = rel_err(out0["energy"].cpu(), ref["energy"]), rel_err(out0["forces"].detach().cpu(), ref["forces"])
    try:
        tabulate.tabulate_filters(model.representation, 512)
        _lib.profile_enable(True); _lib.profile_report()
        out = model(dict(inp))
        tags = _lib.profile_report()
    finally:
        _lib.profile_enable(False)
        tabulate.clear_filter_tables()
    assert "cfconv_tab_fwd" in tags and ("cfconv_tab_bwd" in tags or "cfconv_tab_bwd_geom" in tags), tags
    assert not any(t.startswith(("cfconv_fwd", "cfconv_bwd")) for t in tags), tags
    e1, f1 = rel_err(out["energy"].cpu(), ref["energy"]), rel_err(out["forces"].detach().cpu(), ref["forces"])
    print("contract path: energy %.2e forces %.2e | tabulated: energy %.2e forces %.2e" % (e0, f0, e1, f1))
    assert e1 < TOL and f1 < TOL
    # and the tables are gone again
    out2 = model(dict(inp))
    assert rel_err(out2["forces"].detach().cpu(), out0["forces"].detach().cpu()) < 2e-6


@pytest.mark.gpu
def test_box_split_backward_repeats_and_matches_the_fp32_matrix_path(dev):
    """Round 6 regression: the split-precision pair backward of the general SchNet driver (k_cfconv_pair_t_sp) once came out of the
    compiler with its two per-pair sums in packed instructions between the f16 matrix instructions, and dL/dr_ij of a few dozen
    REVERSED edges per launch was off by percent, differently in every run (profiles/r06_box_split_glitch.md).  Twelve launches of
    the three-interaction backward on a 3 000-atom box against the fp32 matrix path: every edge within 1e-5 of the largest entry."""
    from schnetpack_amd import _lib, model as M
    b = S.water_box(n_side=10, seed=3)
    rep = O.init_schnet_params(128, 3, 20, 5.0)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model("schnet", 128, 3, 20, 5.0)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    r = m.representation
    inp = M.batch_to_inputs(b, dev)
    R = inp["_positions"]
    r_ij = (R[inp["_idx_j"]] - R[inp["_idx_i"]] + inp["_offsets"]).contiguous()
    x0 = r.embedding(inp["_atomic_numbers"]).detach()
    ws = r.interaction_weights()
    kind, p0, p1 = r.radial_basis.kernel_params()
    gx = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))

    def call():
        x, saved, scratch = torch.ops.spk_hip.schnet_forward(x0, r_ij, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True)
        gr, _ = torch.ops.spk_hip.schnet_backward(gx, r_ij, saved, scratch, inp["_idx_i"], inp["_idx_j"], ws, 128, kind, p0, p1, 5.0, True, False)
        torch.cuda.synchronize()
        return gr.detach().clone()

    before = _lib.get_split()
    try:
        _lib.set_split(0)
        ref = call()
        scale = float(ref.abs().max())
        _lib.set_split(1)
        for rep_i in range(12):
            worst = float((call() - ref).abs().max()) / scale
            assert worst < 1e-5, "launch %d: dL/dr_ij off by %.2e of the largest entry" % (rep_i, worst)
    finally:
        _lib.set_split(before)
