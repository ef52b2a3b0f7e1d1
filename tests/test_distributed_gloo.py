"""World-size-2 CPU (gloo) tests of the N>1 host path: frame sharding covers the batch exactly
and the single-bucket gradient all-reduce averages like DDP."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from schnetpack_amd.parallel import FlatGradAllReduce, gather_sharded_results, shard_frames
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.zeros(4))  # never receives a gradient on rank 1
    params = list(lin.parameters()) + [extra]
    x = torch.arange(10.0).view(2, 5) * (rank + 1)
    loss = lin(x).sum() + (extra.sum() if rank == 0 else 0.0)
    loss.backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    FlatGradAllReduce(params)()
    lo, hi = shard_frames(7, rank, world)
    mine = torch.arange(lo, hi, dtype=torch.float32)
    allv = gather_sharded_results(mine)
    q.put((rank, [g.tolist() for g in local], [p.grad.tolist() for p in params], allv.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, local, reduced, allv = q.get(timeout=120)
        res[rank] = (local, reduced, allv)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = torch.tensor
    for k in range(3):
        mean = (T(res[0][0][k]) + T(res[1][0][k])) / 2
        assert torch.allclose(T(res[0][1][k]), mean) and torch.allclose(T(res[1][1][k]), mean)
    assert res[0][2] == list(range(7)) and res[1][2] == list(range(7))


def _train_worker(rank, world, port, q):
    """Three AdamW steps of a small MLP on the rank's half of a batch with the bucket-view reducer."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from schnetpack_amd.parallel import FlatGradAllReduce, shard_frames
    net, data, target = _toy_problem()
    lo, hi = shard_frames(data.shape[0], rank, world)
    red = FlatGradAllReduce(net.parameters(), as_views=True)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    for step in range(3):
        red.zero()
        if step == 1:
            opt.zero_grad(set_to_none=True)        # a caller that drops the views: must be re-bound
        loss = ((net(data[lo:hi]) - target[lo:hi]) ** 2).mean()
        loss.backward()
        red()
        assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters())
        opt.step()
    q.put((rank, [p.detach().tolist() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def _toy_problem():
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.SiLU(), torch.nn.Linear(8, 2))
    data = torch.randn(8, 6)
    target = torch.randn(8, 2)
    return net, data, target


def test_bucket_view_training_world2_equals_single_process():
    """Equal shards + mean loss: the rank-averaged gradient IS the full-batch gradient, so two
    ranks must walk exactly the trajectory of one process on the whole batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net, data, target = _toy_problem()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        ((net(data) - target) ** 2).mean().backward()
        opt.step()
    for k, p in enumerate(net.parameters()):
        for r in range(2):
            assert torch.allclose(torch.tensor(res[r][k]), p.detach(), atol=1e-6), (r, k)
    assert res[0] == res[1]


def _ring_worker(rank, world, port, q):
    """Bead-parallel ring-polymer main step: 4 beads over 2 ranks, one all-gather, each rank evaluates its
    own beads (the compute function is injected: the HIP kernel cannot run in this CPU test)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from schnetpack_amd.md import MDState, RingPolymer
    Q, P, M = _ring_problem()

    def compute(q_all, p_all, masses, A, bead0, n_local):
        A = A.double()
        m = masses.reshape(1, -1, 1).double()
        pn = torch.einsum("bn,nak->bak", A[0], p_all.double()) + m * torch.einsum("bn,nak->bak", A[1], q_all.double())
        qn = torch.einsum("bn,nak->bak", A[2], p_all.double()) / m + torch.einsum("bn,nak->bak", A[3], q_all.double())
        return qn[bead0:bead0 + n_local].float(), pn[bead0:bead0 + n_local].float()

    rp = RingPolymer(5e-4, 4, 300.0, omega=55.0, group=dist.group.WORLD, compute_fn=compute)
    lo, hi = 2 * rank, 2 * rank + 2
    st = MDState(Q[lo:hi].clone(), P[lo:hi].clone(), M)
    rp.main_step(st)
    q.put((rank, st.positions.tolist(), st.momenta.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _ring_problem():
    g = torch.Generator().manual_seed(8)
    return torch.randn(4, 6, 3, generator=g), torch.randn(4, 6, 3, generator=g), torch.rand(1, 6, 1, generator=g) * 10 + 1


def test_bead_parallel_ring_polymer_world2_equals_oracle():
    """The sharded step (all-gather + local bead rows of the folded matrices) equals the oracle's
    transform / propagate / back-transform of all beads (md/integrators.py:204-229)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import md_oracle as MDO
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, qq, pp = q.get(timeout=60)
        res[rank] = (torch.tensor(qq), torch.tensor(pp))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Q, P, M = _ring_problem()
    C = MDO.normal_mode_matrix(4)
    _, prop = MDO.ring_polymer_propagator(4, 55.0, 5e-4)
    q2, p2 = MDO.ring_polymer_main_step(Q.double(), P.double(), M.double(), C, prop)
    got_q = torch.cat([res[0][0], res[1][0]])
    got_p = torch.cat([res[0][1], res[1][1]])
    assert torch.allclose(got_q.double(), q2, atol=1e-5) and torch.allclose(got_p.double(), p2, atol=1e-4)


def _pile_worker(rank, world, port, q):
    """Bead-parallel PILE-L thermostat: 4 beads over 2 ranks, ONE all-gather of the momenta per application, the noise
    regenerated on every rank from the counter alone (the compute function is the host restatement of the HIP kernel:
    same Philox stream, same folded matrices)."""
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import md_oracle as MDO
    from schnetpack_amd.md import MDState, PILELocalThermostat, RingPolymer
    Q, P, M = _ring_problem()

    def compute(p_all, masses, Mx, noise_scale, seed, step, step_dev, which, bead0, n_local, out=None):
        B, n = p_all.shape[0], p_all.shape[1]
        xi = MDO.pile_noise(B, n, seed, step, which)
        det = torch.einsum("bn,nak->bak", Mx[0].double(), p_all.double())
        noi = torch.einsum("bk,kat->bat", Mx[1].double(), xi)
        res = det + torch.sqrt(masses.reshape(1, -1, 1).double()) * noise_scale * noi
        return res[bead0:bead0 + n_local].float()

    rp = RingPolymer(5e-4, 4, 300.0, omega=55.0, group=dist.group.WORLD, compute_fn=lambda *a, **k: None)
    th = PILELocalThermostat(300.0, 100.0, seed=99, group=dist.group.WORLD, compute_fn=compute).init(rp)
    lo, hi = 2 * rank, 2 * rank + 2
    st = MDState(Q[lo:hi].clone(), P[lo:hi].clone(), M)
    th.apply(st, step=5, which=0)
    th.apply(st, step=5, which=1)
    q.put((rank, st.momenta.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_bead_parallel_pile_thermostat_world2_equals_single_process_oracle():
    """Two ranks x two beads == all four beads in one process through the reference's formula
    (thermostats_rpmd.py:102-119) with the same counter-based noise: no exchange beyond the momenta all-gather."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import md_oracle as MDO
    from schnetpack_amd import md as MD
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pile_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, pp = q.get(timeout=60)
        res[rank] = torch.tensor(pp)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Q, P, M = _ring_problem()
    C = MDO.normal_mode_matrix(4)
    c1, c2 = MDO.pile_coefficients(4, 55.0, 5e-4, 0.1)
    kT = MD.KB_MD * 4 * 300.0
    p1 = MDO.pile_apply(P.double(), M.double(), C, c1, c2, kT, MDO.pile_noise(4, 6, 99, 5, 0))
    p2 = MDO.pile_apply(p1, M.double(), C, c1, c2, kT, MDO.pile_noise(4, 6, 99, 5, 1))
    got = torch.cat([res[0], res[1]])
    assert torch.allclose(got.double(), p2, rtol=1e-5, atol=1e-5 * float(p2.abs().max()))


# ------------------------------------------------------------------------------------------------ configs[4] as stated: 8 beads, one per rank
def _bead8_problem():
    g = torch.Generator().manual_seed(21)
    return torch.randn(8, 5, 3, generator=g), torch.randn(8, 5, 3, generator=g), torch.rand(1, 5, 1, generator=g) * 10 + 1


def _toy_forces(q):
    """A bead-local force field (harmonic wells + a quartic term): stands in for the force call of a bead, which couples nothing
    across beads (md/calculators/base_calculator.py:166-183 folds the beads into the batch dimension)."""
    return -q - 0.1 * q ** 3


def _bead8_worker(rank, world, port, q, exchange):
    """One NVT ring-polymer step of configs[4] -- PILE-L, half kick, bead mixing, force call, half kick, PILE-L
    (md/simulator.py:126-150, md/integrators.py:204-229, md/simulation_hooks/thermostats_rpmd.py:102-119) -- with ONE bead per rank.
    exchange == "state": a rank holds only its bead; the thermostat applications and the bead mixing all-gather (3 collectives).
    exchange == "forces": every rank integrates all 8 beads (counter-based noise keeps the replicas identical) and evaluates the
    forces of its own bead only; the forces are all-gathered (1 collective).  The compute functions are the host restatements of
    the HIP kernels (the kernels cannot run in this CPU test); the exchange logic is the product's."""
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import md_oracle as MDO
    from schnetpack_amd.md import MDState, PILELocalThermostat, RingPolymer
    Q, P, M = _bead8_problem()
    B, dt = 8, 2e-4
    count = {"n": 0}
    real_gather = dist.all_gather_into_tensor

    def counting_gather(*a, **k):
        count["n"] += 1
        return real_gather(*a, **k)
    dist.all_gather_into_tensor = counting_gather

    def ring_compute(q_all, p_all, masses, A, bead0, n_local):
        A = A.double()
        m = masses.reshape(1, -1, 1).double()
        pn = torch.einsum("bn,nak->bak", A[0], p_all.double()) + m * torch.einsum("bn,nak->bak", A[1], q_all.double())
        qn = torch.einsum("bn,nak->bak", A[2], p_all.double()) / m + torch.einsum("bn,nak->bak", A[3], q_all.double())
        return qn[bead0:bead0 + n_local].float(), pn[bead0:bead0 + n_local].float()

    def pile_compute(p_all, masses, Mx, noise_scale, seed, step, step_dev, which, bead0, n_local, out=None):
        xi = MDO.pile_noise(p_all.shape[0], p_all.shape[1], seed, step, which)
        det = torch.einsum("bn,nak->bak", Mx[0].double(), p_all.double())
        noi = torch.einsum("bk,kat->bat", Mx[1].double(), xi)
        res = det + torch.sqrt(masses.reshape(1, -1, 1).double()) * noise_scale * noi
        return res[bead0:bead0 + n_local].float()

    grp = dist.group.WORLD if exchange == "state" else None
    rp = RingPolymer(dt, B, 300.0, omega=40.0, group=grp, compute_fn=ring_compute)
    th = PILELocalThermostat(300.0, 100.0, seed=7, group=grp, compute_fn=pile_compute).init(rp)
    lo, hi = (rank, rank + 1) if exchange == "state" else (0, B)
    st = MDState(Q[lo:hi].clone(), P[lo:hi].clone(), M)
    F = _toy_forces(st.positions)
    th.apply(st, step=3, which=0)
    st.momenta += 0.5 * dt * F
    rp.main_step(st)
    if exchange == "state":
        F = _toy_forces(st.positions)
    else:                                   # the sharded force call: own bead only, then ONE all-gather of [1, N, 3] per rank
        mine = _toy_forces(st.positions[rank:rank + 1]).contiguous()
        F = torch.empty_like(st.positions)
        dist.all_gather_into_tensor(F.view(-1), mine.view(-1))
    st.momenta += 0.5 * dt * F
    th.apply(st, step=3, which=1)
    q.put((rank, st.positions.tolist(), st.momenta.tolist(), count["n"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["state", "forces"])
def test_configs4_eight_beads_one_per_rank_world8(exchange):
    """BASELINE configs[4] -- 8 PIMD beads, one bead per GPU -- on 8 gloo ranks: both exchange schemes reproduce the single-process
    NVT step of the oracle (reference formulae in float64, same Philox stream) and issue exactly the collectives DESIGN.md section 6
    states: 3 all-gathers per step (state) or 1 (forces)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import md_oracle as MDO
    from schnetpack_amd import md as MD
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bead8_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, qq, pp, ncoll = q.get(timeout=180)
        res[rank] = (torch.tensor(qq), torch.tensor(pp), ncoll)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process oracle of the same step
    Q, P, M = _bead8_problem()
    B, dt = 8, 2e-4
    C = MDO.normal_mode_matrix(B)
    _, prop = MDO.ring_polymer_propagator(B, 40.0, dt)
    c1, c2 = MDO.pile_coefficients(B, 40.0, dt, 0.1)
    kT = MD.KB_MD * B * 300.0
    qd, pd, md_ = Q.double(), P.double(), M.double()
    F = _toy_forces(qd)
    pd = MDO.pile_apply(pd, md_, C, c1, c2, kT, MDO.pile_noise(B, 5, 7, 3, 0))
    pd = pd + 0.5 * dt * F
    qd, pd = MDO.ring_polymer_main_step(qd, pd, md_, C, prop)
    pd = pd + 0.5 * dt * _toy_forces(qd)
    pd = MDO.pile_apply(pd, md_, C, c1, c2, kT, MDO.pile_noise(B, 5, 7, 3, 1))
    if exchange == "state":
        got_q = torch.cat([res[r][0] for r in range(world)])
        got_p = torch.cat([res[r][1] for r in range(world)])
        assert all(res[r][2] == 3 for r in range(world)), [res[r][2] for r in range(world)]
    else:
        got_q, got_p = res[0][0], res[0][1]
        for r in range(1, world):              # the replicas stay bit-identical
            assert torch.equal(res[r][0], got_q) and torch.equal(res[r][1], got_p)
        assert all(res[r][2] == 1 for r in range(world)), [res[r][2] for r in range(world)]
    assert torch.allclose(got_q.double(), qd, atol=1e-5) and torch.allclose(got_p.double(), pd, rtol=1e-5, atol=1e-5 * float(pd.abs().max()))


# ------------------------------------------------------------------------------------------------ GraphedTrainStep: the two-graph split
class _FakeGraph:
    """Stands in for torch.cuda.CUDAGraph on the build box: 'capture' records which of the step's pieces were issued inside it
    (nothing executes, as under a real capture); replay runs them (the control flow is the product's, the graphs are not)."""
    log = None           # the run log of the process
    current = None

    def __init__(self):
        self.pieces = []

    def pool(self):
        return None

    def replay(self):
        for name, fn in self.pieces:
            _FakeGraph.log.append(name)
            fn()


class _fake_graph_ctx:
    def __init__(self, g, pool=None):
        self.g = g

    def __enter__(self):
        _FakeGraph.current = self.g

    def __exit__(self, *exc):
        _FakeGraph.current = None
        return False


class _NullLists:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _graphed_step_worker(rank, world, port, q):
    """GraphedTrainStep.step() with world size `world`: capture decision (one graph / backward + optimizer graphs), the order
    backward-graph -> flat-bucket all-reduce -> optimizer-graph, and the averaged update.  The model's forward / backward and the
    optimizer are stand-ins (a quadratic loss with rank-dependent data, plain SGD): what is under test is train.py's control flow."""
    import torch.distributed as dist
    from schnetpack_amd import train as T
    from schnetpack_amd.parallel import FlatGradAllReduce
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.zeros(4))
    target = torch.arange(4.0) * (rank + 1)          # rank-dependent data: the averaged gradient differs from either local one
    st = object.__new__(T.GraphedTrainStep)
    st.dev, st.group, st.use_graph, st.warmup_steps, st.n_steps = torch.device("cpu"), None, True, 0, 0
    st.g_bwd = st.g_opt = None
    st.lists = _NullLists()
    st.loss = torch.zeros(())
    st.reducer = FlatGradAllReduce([w], as_views=True)
    _FakeGraph.log = log = []

    def forward_backward():
        def body():
            st.reducer.release()
            loss = 0.5 * ((w - target) ** 2).sum()
            loss.backward()
            st.reducer.pack()
            st.loss.copy_(loss.detach())
        if _FakeGraph.current is not None:           # under capture nothing executes: the work is recorded
            _FakeGraph.current.pieces.append(("fb", body))
        else:
            body()

    class Opt:
        def step(self_inner):
            def body():
                with torch.no_grad():
                    w.sub_(0.5 * w.grad)
            if _FakeGraph.current is not None:
                _FakeGraph.current.pieces.append(("opt", body))
            else:
                body()
    st._forward_backward = forward_backward
    st.opt = Opt()
    real_reducer = st.reducer

    class LoggedReducer:
        def __call__(self_inner, group=None):
            log.append("reduce")
            return real_reducer(group)

        def __getattr__(self_inner, name):
            return getattr(real_reducer, name)
    st.reducer = LoggedReducer()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.CUDAGraph = _FakeGraph
    torch.cuda.graph = _fake_graph_ctx
    torch.ops.spk_hip.weights_changed = lambda: None if False else None
    T.torch.ops.spk_hip.weights_changed()          # (exists on the host: no tensors involved)
    st.step()                                      # capture + first replay
    captured = ([n for n, _ in st.g_bwd.pieces], None if st.g_opt is None else [n for n, _ in st.g_opt.pieces])
    del log[:]
    st.step()
    q.put((rank, captured, list(log), w.detach().tolist()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_graphed_train_step_two_graph_split(world):
    """Round-3 review: `GraphedTrainStep._capture` decides at capture time whether the optimizer goes into the backward graph (one
    rank) or into a second graph behind the gradient all-reduce (several ranks) -- exercised here with world size 1 and 2 on gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphed_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, captured, log, wv = q.get(timeout=180)
        res[rank] = (captured, log, wv)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        captured, log, wv = res[rank]
        if world == 1:
            assert captured == (["fb", "opt"], None) and log == ["fb", "opt"]
        else:
            assert captured == (["fb"], ["opt"]) and log == ["fb", "reduce", "opt"]
    # the trajectory: SGD with lr 0.5 on 0.5 |w - t|^2, two applications of the update (two replays), with the rank-AVERAGED
    # target when there are two ranks: w <- w + 0.5 (t_mean - w)
    t_mean = torch.arange(4.0) * (1.5 if world == 2 else 1.0)
    want = torch.zeros(4)
    for _ in range(2):
        want = want + 0.5 * (t_mean - want)
    for rank in range(world):
        assert torch.allclose(torch.tensor(res[rank][2]), want, atol=1e-6), (rank, res[rank][2], want.tolist())
