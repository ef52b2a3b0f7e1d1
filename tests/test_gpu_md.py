"""GPU parity of the MD-step kernels and the graphed force call."""
import math

import pytest
import torch

from conftest import load_npz, rel_err
from oracle import md_oracle as MDO
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda", 0)


def test_velocity_verlet_steps_match_oracle(dev):
    """md/integrators.py:59-70, :97-110 -- separate steps, the fused pass, and the skin flag."""
    from schnetpack_amd.md import MDState, VelocityVerlet
    g = torch.Generator().manual_seed(0)
    n_rep, n = 3, 1000
    R, p, F = (torch.randn(n_rep, n, 3, generator=g) for _ in range(3))
    m = torch.rand(1, n, 1, generator=g) * 15 + 1
    dt = 0.37
    vv = VelocityVerlet(dt)
    st = MDState(R.to(dev).clone(), p.to(dev).clone(), m.to(dev), F.to(dev))
    vv.half_step(st)
    p1 = MDO.half_step(p.double(), F.double(), dt)
    assert rel_err(st.momenta.cpu(), p1) < 1e-6
    vv.main_step(st)
    R1 = MDO.verlet_main_step(R.double(), p1, m.double(), dt)
    assert rel_err(st.positions.cpu(), R1) < 1e-6 and rel_err(st.momenta.cpu(), p1) < 1e-6
    # fused first half + main step with the skin criterion
    st2 = MDState(R.to(dev).clone(), p.to(dev).clone(), m.to(dev), F.to(dev))
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    disp = (R1 - R.double()).norm(dim=-1).max().item()
    vv.first_half_and_main_step(st2, True, R.to(dev).reshape(-1, 3).contiguous(), 1.01 * disp, flag)
    assert rel_err(st2.positions.cpu(), R1) < 1e-6 and torch.equal(st2.momenta, st.momenta)
    assert int(flag[0].item()) == 0
    # flag[1]: bits of the largest squared one-step displacement
    step2 = flag[1:].view(torch.float32).item()
    assert abs(step2 ** 0.5 - disp) < 1e-5 * disp
    st3 = MDState(R.to(dev).clone(), p.to(dev).clone(), m.to(dev), F.to(dev))
    vv.first_half_and_main_step(st3, True, R.to(dev).reshape(-1, 3).contiguous(), 0.99 * disp, flag)
    assert int(flag[0].item()) == 1


@pytest.mark.parametrize("nb", [1, 2, 4, 5, 8])
def test_ring_polymer_main_step_matches_reference_vectors(dev, nb):
    """spk_md_ring_polymer_step_f32 against outputs of the reference's RingPolymer._main_step
    (tests/golden/md_ring_polymer.npz) and, at size, against the oracle; bead sub-ranges (what a rank of
    a bead-parallel run computes) equal slices of the full result."""
    from schnetpack_amd.md import MDState, RingPolymer
    g = load_npz("md_ring_polymer.npz")
    t = "b%d_" % nb
    q, p, m = (torch.from_numpy(g[t + k]) for k in ("q", "p", "m"))
    rp = RingPolymer(float(g[t + "dt"]), nb, 300.0, omega=float(g[t + "omega"]))
    st = MDState(q.float().to(dev), p.float().to(dev), m.float().to(dev))
    rp.main_step(st)
    assert rel_err(st.positions.cpu(), torch.from_numpy(g[t + "q_out"])) < TOL
    assert rel_err(st.momenta.cpu(), torch.from_numpy(g[t + "p_out"])) < TOL
    # larger system, sub-ranges
    gen = torch.Generator().manual_seed(nb)
    n = 3000
    Q, Pm = torch.randn(nb, n, 3, generator=gen), torch.randn(nb, n, 3, generator=gen)
    M = torch.rand(1, n, 1, generator=gen) * 15 + 1
    C = MDO.normal_mode_matrix(nb)
    _, prop = MDO.ring_polymer_propagator(nb, rp.omega, rp.time_step)
    q2, p2 = MDO.ring_polymer_main_step(Q.double(), Pm.double(), M.double(), C, prop)
    from schnetpack_amd.md import _ring_polymer_hip
    for lo, hi in {(0, nb), (nb // 2, nb), (0, max(1, nb // 2))}:
        qo, po = _ring_polymer_hip(Q.to(dev), Pm.to(dev), M.to(dev), rp.A.to(dev), lo, hi - lo)
        assert rel_err(qo.cpu(), q2[lo:hi]) < TOL and rel_err(po.cpu(), p2[lo:hi]) < TOL


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_graphed_force_call_equals_eager(dev, kind):
    """Replay of the captured force call == the eager call, for moving positions on a fixed list; a new
    list re-captures."""
    from schnetpack_amd import model as M
    from schnetpack_amd.forcecall import GraphedForceCall
    rep_p = O.init_schnet_params() if kind == "schnet" else O.init_painn_params()
    head_p = O.init_atomwise_params(128, seed=1)
    model = M.build_model(kind)
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).eval()
    b = S.molecule_batch("aspirin", 6, seed=3)
    inp = M.batch_to_inputs(b, dev)
    fc = GraphedForceCall(model)
    gen = torch.Generator().manual_seed(1)
    for it in range(3):
        call = dict(inp)
        call["_positions"] = inp["_positions"].detach() + 0.02 * it * torch.randn(inp["_positions"].shape, generator=gen).to(dev)
        got = fc(call)
        want = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in call.items()})
        assert rel_err(got["forces"].cpu(), want["forces"].detach().cpu()) < 1e-6
        assert rel_err(got["energy"].cpu(), want["energy"].detach().cpu()) < 1e-6
    assert fc.n_captures == 1
    b2 = S.molecule_batch("aspirin", 4, seed=9)
    got = fc(M.batch_to_inputs(b2, dev))
    ref = O.energy_and_forces(kind, rep_p, head_p, b2, 3)
    assert fc.n_captures == 2 and rel_err(got["forces"].cpu(), ref["forces"]) < TOL


@pytest.mark.parametrize("kind", ["schnet", "painn"])
@pytest.mark.parametrize("n_mol", [6, 40, 256])
def test_graph_replays_stay_correct(dev, kind, n_mol):
    """Regression: every replay of the captured force call (not only the first) reproduces the eager
    result.  (hipMemsetAsync captured as a memset node left its target un-cleared from the second replay on;
    the library clears buffers with a kernel now -- spk_zero_async.)"""
    from schnetpack_amd import model as M
    from schnetpack_amd.forcecall import GraphedForceCall
    rep_p = O.init_schnet_params() if kind == "schnet" else O.init_painn_params()
    head_p = O.init_atomwise_params(128, seed=1)
    model = M.build_model(kind)
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).eval()
    b = S.molecule_batch("aspirin", n_mol, seed=3)
    inp = M.batch_to_inputs(b, dev)
    want = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()})
    we, wf = want["energy"].detach().cpu(), want["forces"].detach().cpu()
    fc = GraphedForceCall(model)
    for it in range(5):
        got = fc(dict(inp)) if it % 2 == 0 else fc.replay()
        assert rel_err(got["forces"].cpu(), wf) < 1e-5, (it, "forces")
        assert rel_err(got["energy"].cpu(), we) < 1e-5, (it, "energy")
    assert fc.n_captures == 1


@pytest.mark.parametrize("complete_list", [False, "auto"])
def test_nve_loop_conserves_energy_and_follows_oracle_trajectory(dev, complete_list):
    """End to end: device neighbour list with skin (or, "auto" for these small isolated molecules, the complete intramolecular
    list that never needs a rebuild and runs without host synchronisation) + graphed SchNet force call + fused Verlet kernels.
    (i) the first 10 steps follow a float64 CPU integration of the ORACLE forces; (ii) total energy is
    conserved over 400 steps while the list is rebuilt several times (forces are the exact gradient of the
    energy, the list never misses a pair)."""
    from schnetpack_amd import model as M
    from schnetpack_amd.md import NVESimulation
    rep_p, head_p = O.init_schnet_params(), O.init_atomwise_params(128, seed=1)
    model = M.build_model("schnet")
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).eval()
    b = S.molecule_batch("aspirin", 4, seed=1, jitter=0.02)
    inp = M.batch_to_inputs(b, dev)
    inp["_n_atoms"] = torch.full((4,), 21, device=dev)
    masses = torch.where(b["Z"] == 1, 1.008, torch.where(b["Z"] == 6, 12.011, 15.999))
    dt = 0.02
    sim = NVESimulation(model, inp, masses.to(dev), dt, cutoff=5.0, cutoff_shell=0.3, complete_list=complete_list)
    assert sim._complete == (complete_list == "auto")
    if sim._complete:
        assert int(sim._lists["_idx_i"].shape[0]) == 4 * 21 * 20
    g = torch.Generator().manual_seed(0)
    p0 = 0.3 * torch.randn(b["R"].shape, generator=g) * masses[:, None].sqrt()
    sim.state.momenta.copy_(p0.to(dev).unsqueeze(0))
    e0 = sim.total_energy()

    # float64 oracle trajectory (full list of each molecule: cutoff 5 A covers an aspirin molecule's pairs within 5 A)
    def oracle_forces(R):
        from oracle import nbl_oracle as NB
        i, j, _, off = NB.batch_neighbor_list(R.float(), b["idx_m"], None, None, 5.0)
        bb = dict(b, R=R, idx_i=i, idx_j=j, offsets=off.double())
        return O.energy_and_forces("schnet", rep_p, head_p, bb, 3, dtype=torch.float64)["forces"]

    R, p, m = b["R"].double(), p0.double(), masses.double()[:, None]
    F = oracle_forces(R)
    for _ in range(10):
        p = p + 0.5 * dt * F
        R = R + dt * p / m
        F = oracle_forces(R)
        p = p + 0.5 * dt * F
    sim.step(10)
    assert rel_err(sim.state.positions[0].cpu(), R) < 1e-5
    assert rel_err(sim.state.momenta[0].cpu(), p) < 1e-4
    sim.step(390)
    ke = float(sim.kinetic_energy())
    drift = abs(sim.total_energy() - e0)
    assert drift < 2e-3 * ke, (drift, ke, e0)
    if sim._complete:
        assert sim.nl.n_builds == 1 and sim.n_captures == 1        # one list, one captured graph for the whole run
    else:
        assert sim.nl.n_builds >= 2, sim.nl.n_builds


def test_rpmd_loop_conserves_ring_polymer_energy_and_follows_oracle(dev):
    """Ring-polymer MD, 4 beads folded into the batch (md/integrators.py:113-229): (i) 8 steps follow a float64
    CPU integration with the ORACLE's half step / ring-polymer main step and oracle forces per bead;
    (ii) the ring-polymer Hamiltonian (kinetic + sum of bead potentials + springs) is conserved over 300 steps."""
    from oracle import nbl_oracle as NB
    from schnetpack_amd import model as M
    from schnetpack_amd.md import RPMDSimulation
    rep_p, head_p = O.init_schnet_params(), O.init_atomwise_params(128, seed=1)
    model = M.build_model("schnet")
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).eval()
    b = S.molecule_batch("aspirin", 2, seed=2, jitter=0.02)
    N, B, dt, omega = b["Z"].shape[0], 4, 0.02, 3.0
    inp = M.batch_to_inputs(b, dev)
    inp["_n_atoms"] = torch.full((2,), 21, device=dev)
    masses = torch.where(b["Z"] == 1, 1.008, torch.where(b["Z"] == 6, 12.011, 15.999))
    sim = RPMDSimulation(model, inp, masses.to(dev), dt, B, cutoff=5.0, omega=omega, cutoff_shell=0.4)
    g = torch.Generator().manual_seed(3)
    q0 = b["R"][None].repeat(B, 1, 1) + 0.03 * torch.randn(B, N, 3, generator=g)
    p0 = 0.2 * torch.randn(B, N, 3, generator=g) * masses[None, :, None].sqrt()
    sim.state.positions.copy_(q0.to(dev))
    sim.state.momenta.copy_(p0.to(dev))
    sim._rebuild(True)              # list, forces and graph for the new bead positions
    sim._force_eval()
    e0 = sim.total_energy()

    def oracle_forces(q):           # per bead, exact lists
        out = []
        for k in range(B):
            i, j, _, off = NB.batch_neighbor_list(q[k].float(), b["idx_m"], None, None, 5.0)
            bb = dict(b, R=q[k], idx_i=i, idx_j=j, offsets=off.double())
            out.append(O.energy_and_forces("schnet", rep_p, head_p, bb, 3, dtype=torch.float64)["forces"])
        return torch.stack(out)

    C = MDO.normal_mode_matrix(B)
    _, prop = MDO.ring_polymer_propagator(B, omega, dt)
    q, p, m = q0.double(), p0.double(), masses.double()[None, :, None]
    F = oracle_forces(q)
    for _ in range(8):
        p = MDO.half_step(p, F, dt)
        q, p = MDO.ring_polymer_main_step(q, p, m, C, prop)
        F = oracle_forces(q)
        p = MDO.half_step(p, F, dt)
    sim.step(8)
    assert rel_err(sim.state.positions.cpu(), q) < 1e-5
    assert rel_err(sim.state.momenta.cpu(), p) < 1e-4
    sim.step(292)
    ke = float(sim.kinetic_energy())
    assert abs(sim.total_energy() - e0) < 2e-3 * ke, (sim.total_energy(), e0, ke)


def test_nve_periodic_water_box_painn_conserves_energy(dev):
    """Periodic 192-atom water box, PaiNN, device cell list with cell offsets and a skin, graph-replayed steps:
    total energy is conserved over 300 steps while atoms cross the skin threshold (several rebuilds) -- exercises
    the offsets of the periodic list through the fused PaiNN kernels (incl. the live-edge mask) in dynamics."""
    from schnetpack_amd import model as M
    from schnetpack_amd.md import NVESimulation
    rep_p, head_p = O.init_painn_params(), O.init_atomwise_params(128, seed=1)
    model = M.build_model("painn")
    M.load_reference_params(model, rep_p, head_p)
    model = model.to(dev).eval()
    b = S.water_box(n_side=4, seed=7)
    inp = M.batch_to_inputs(b, dev)
    inp["_n_atoms"] = torch.tensor([b["Z"].shape[0]], device=dev)
    inp["_cell"] = b["cell"].reshape(1, 3, 3).to(dev)
    inp["_pbc"] = torch.tensor([True, True, True], device=dev)
    masses = torch.where(b["Z"] == 1, 1.008, 15.999)
    sim = NVESimulation(model, inp, masses.to(dev), 0.01, cutoff=5.0, cutoff_shell=0.3)
    g = torch.Generator().manual_seed(1)
    sim.state.momenta.copy_((0.5 * torch.randn(b["R"].shape, generator=g) * masses[:, None].sqrt()).to(dev).unsqueeze(0))
    e0 = sim.total_energy()
    sim.step(300)
    ke = float(sim.kinetic_energy())
    assert abs(sim.total_energy() - e0) < 2e-3 * ke, (sim.total_energy(), e0, ke)
    assert sim.nl.n_builds >= 2
    # the list in use equals a fresh device search with cutoff + skin at the reference positions
    assert sim._lists["_idx_i"].shape[0] > 0


# ----------------------------------------------------------------------------- PILE-L thermostat (row f3)
@pytest.mark.parametrize("n_beads,n_local,bead0", [(4, 4, 0), (8, 8, 0), (8, 2, 4), (5, 5, 0), (16, 10, 3)])
def test_pile_thermostat_kernel_matches_oracle_with_the_same_noise(dev, n_beads, n_local, bead0):
    """spk_md_pile_f32 against the reference's formula (thermostats_rpmd.py:102-119, restated in oracle/md_oracle.py) fed the
    SAME normal-mode noise -- the counter-based stream restated on the host (Philox-4x32-10 + Box-Muller)."""
    from oracle import md_oracle as MDO
    from schnetpack_amd import md as MD
    g = torch.Generator().manual_seed(5)
    n_atoms, omega, dt, tau, T = 37, 55.0, 5e-4, 0.1, 300.0
    p = torch.randn(n_beads, n_atoms, 3, generator=g)
    masses = (torch.rand(1, n_atoms, 1, generator=g) * 15 + 1)
    M = MD.pile_matrices(n_beads, omega, dt, tau)
    scale = math.sqrt(MD.KB_MD * n_beads * T)
    seed, step, which = 0x1234567ABCDEF, 41, 1
    got = MD._pile_hip(p.to(dev), masses.to(dev), M.to(dev), scale, seed, step, None, which, bead0, n_local).cpu()
    C = MDO.normal_mode_matrix(n_beads)
    c1, c2 = MDO.pile_coefficients(n_beads, omega, dt, tau)
    xi = MDO.pile_noise(n_beads, n_atoms, seed, step, which)
    ref = MDO.pile_apply(p.double(), masses.double(), C, c1, c2, MD.KB_MD * n_beads * T, xi)[bead0:bead0 + n_local]
    assert torch.allclose(got.double(), ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    # a device-resident step counter gives the same stream as the host value
    stepc = torch.tensor([step], dtype=torch.int64, device=dev)
    got2 = MD._pile_hip(p.to(dev), masses.to(dev), M.to(dev), scale, seed, 0, stepc, which, bead0, n_local).cpu()
    assert torch.equal(got, got2)


def test_pile_thermostat_noise_statistics_and_equilibrium(dev):
    """The stochastic part alone: normal-mode momenta of a free ring polymer under repeated application reach
    <p_k^2> = m kB n T for every thermostatted mode (the fixed point of p' = c1 p + sqrt(m kB n T) c2 xi)."""
    from oracle import md_oracle as MDO
    from schnetpack_amd import md as MD
    n_beads, n_atoms, omega, dt, tau, T = 4, 4096, 55.0, 5e-3, 0.02, 300.0
    M = MD.pile_matrices(n_beads, omega, dt, tau).to(dev)
    masses = torch.full((1, n_atoms, 1), 12.0, device=dev)
    p = torch.zeros(n_beads, n_atoms, 3, device=dev)
    scale = math.sqrt(MD.KB_MD * n_beads * T)
    for step in range(400):
        p = MD._pile_hip(p, masses, M, scale, 7, step, None, 0, 0, n_beads)
    C = MDO.normal_mode_matrix(n_beads).float().to(dev)
    pn = (C @ p.reshape(n_beads, -1)).view(p.shape)
    var = (pn ** 2).mean(dim=(1, 2)).cpu()
    target = 12.0 * MD.KB_MD * n_beads * T
    assert torch.allclose(var, torch.full_like(var, target), rtol=0.05), (var, target)
    # successive applications draw different noise; different seeds too
    a = MD._pile_hip(p, masses, M, scale, 7, 1000, None, 0, 0, n_beads)
    b = MD._pile_hip(p, masses, M, scale, 7, 1001, None, 0, 0, n_beads)
    c = MD._pile_hip(p, masses, M, scale, 8, 1000, None, 0, 0, n_beads)
    d = MD._pile_hip(p, masses, M, scale, 7, 1000, None, 1, 0, n_beads)
    assert not torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)


def test_rpmd_nvt_loop_with_pile_thermostat_thermalises(dev):
    """RPMDSimulation(thermostat=PILELocalThermostat): the graph-replayed NVT loop (thermostat at step begin and end, device
    step counter) heats a cold ring polymer towards the bath temperature and stays finite."""
    from schnetpack_amd import md as MD, model as M
    b = S.molecule_batch("aspirin", 4, seed=2)
    torch.manual_seed(0)
    model = M.build_model("schnet").to(dev).eval()
    inp = M.batch_to_inputs(b, dev)
    inp["_n_atoms"] = torch.bincount(b["idx_m"], minlength=4).to(dev)
    masses = torch.where(b["Z"] == 1, 1.008, torch.where(b["Z"] == 6, 12.011, 15.999)).to(dev)
    th = MD.PILELocalThermostat(300.0, 10.0, seed=3)        # time constant in fs, like the reference's
    sim = MD.RPMDSimulation(model, inp, masses, 2e-4, 4, cutoff=5.0, omega=30.0, cutoff_shell=2.0, thermostat=th)
    assert float(sim.kinetic_energy()) == 0.0
    sim.step(300)
    ke = float(sim.kinetic_energy())
    n_dof = 3 * 4 * b["Z"].shape[0]
    T_est = 2.0 * ke / (n_dof * MD.KB_MD) / 4          # ring-polymer momenta carry n_beads T
    assert 50.0 < T_est < 900.0, T_est
    assert int(sim._stepc.item()) == 300
