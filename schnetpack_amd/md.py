"""MD steps around the force call: the velocity-Verlet and ring-polymer
integrators of the reference (md/integrators.py:24-229) on fused HIP kernels, and a bead-parallel ring
polymer (one bead per rank, one all-gather per step).  Unit agnostic: ``time_step`` is in the unit system
of the tensors handed in (the reference's MD internal units are kJ/mol, nm, Dalton => ps; 1 fs = 1e-3).

``state`` objects only need ``positions``, ``momenta``, ``forces`` ([n_replicas, n_atoms, 3]) and
``masses`` ([1, n_atoms, 1] or [n_atoms]) attributes -- what ``schnetpack.md.System`` has.
"""
import math
import os
import struct
from typing import Optional

import torch

from . import _lib
from ._lib import check, fptr, lib, stream

__all__ = ["PILELocalThermostat", "pile_matrices", "VelocityVerlet", "RingPolymer", "NVESimulation", "RPMDSimulation", "MDState", "normal_mode_matrix", "ring_polymer_propagator", "ring_polymer_matrices",
           "KB_MD", "HBAR_MD", "FS_MD"]

# reference MD internal units (kJ/mol, nm, Dalton): time unit = 1 ps (units.py:10-40)
FS_MD = 1.0e-3
KB_MD = 8.314462618e-3          # kJ / (mol K)
HBAR_MD = 6.350779923e-2        # kJ / mol * ps


def _flat_masses(masses: torch.Tensor, n_atoms: int) -> torch.Tensor:
    m = masses.reshape(-1)
    if m.numel() != n_atoms:
        raise _lib.SpkHipError("masses: expected %d entries, got %d" % (n_atoms, m.numel()))
    return m.float().contiguous()


class VelocityVerlet:
    """md/integrators.py:24-110.  ``half_step`` / ``main_step`` as in the reference, plus
    ``first_half_and_main_step`` which fuses the two that are adjacent in the MD loop
    (md/simulator.py:126-150) into one pass and evaluates the neighbour-list skin criterion on the way."""

    ring_polymer = False
    pressure_control = False

    def __init__(self, time_step: float):
        self.time_step = float(time_step)

    def half_step(self, state):
        p = state.momenta
        with torch.cuda.device(p.device):
            check(lib().spk_md_half_step_f32(fptr(p), fptr(state.forces.contiguous()), 0.5 * self.time_step, p.numel(), stream()))

    def main_step(self, state):
        self.first_half_and_main_step(state, kick=False)

    def first_half_and_main_step(self, state, kick: bool = True, reference_positions: Optional[torch.Tensor] = None,
                                 max_displacement: float = 0.0, flag: Optional[torch.Tensor] = None):
        R, p = state.positions, state.momenta
        n_rep = R.shape[0] if R.dim() == 3 else 1
        n_atoms = R.numel() // 3
        m = _flat_masses(state.masses, n_atoms // n_rep)
        if n_rep > 1:
            m = m.repeat(n_rep)
        F = state.forces.contiguous() if kick else None
        with torch.cuda.device(R.device):
            check(lib().spk_md_kick_drift_f32(fptr(R), fptr(p), fptr(F), fptr(m), self.time_step, n_atoms,
                                              fptr(reference_positions), float(max_displacement) ** 2,
                                              _lib.iptr(flag, torch.int32) if flag is not None else None, stream()))


def normal_mode_matrix(n_beads: int) -> torch.Tensor:
    """C[k, n] (md/utils/normal_model_transformation.py:38-68), float64."""
    B = n_beads
    n = torch.arange(1, B + 1, dtype=torch.float64)
    C = torch.zeros(B, B, dtype=torch.float64)
    C[0] = 1.0
    for k in range(1, B // 2 + 1):
        C[k] = math.sqrt(2.0) * torch.cos(2.0 * math.pi * k * n / B)
    for k in range(B // 2 + 1, B):
        C[k] = math.sqrt(2.0) * torch.sin(2.0 * math.pi * k * n / B)
    if B % 2 == 0:
        C[B // 2] = torch.where(n.long() % 2 == 0, 1.0, -1.0).double()
    return C / math.sqrt(B)


def ring_polymer_propagator(n_beads: int, omega: float, time_step: float) -> torch.Tensor:
    """[n_beads, 2, 2] free ring-polymer propagator in normal modes (md/integrators.py:152-199)."""
    on = 2.0 * omega * torch.sin(torch.arange(n_beads).float() * math.pi / n_beads)
    odt = on * time_step
    P = torch.zeros(n_beads, 2, 2)
    P[:, 0, 0] = torch.cos(odt)
    P[:, 1, 1] = torch.cos(odt)
    P[:, 0, 1] = -torch.sin(odt) * on
    P[1:, 1, 0] = torch.sin(odt)[1:] / on[1:]
    P[0, 1, 0] = time_step
    return P


def ring_polymer_matrices(n_beads: int, omega: float, time_step: float) -> torch.Tensor:
    """A [4, B, B] = C^T diag(P_ij) C for (ij) = pp, pq, qp, qq: transform, propagate and back-transform
    folded into bead-space matrices (they are linear maps; evaluated in float64, stored float32)."""
    C = normal_mode_matrix(n_beads)
    P = ring_polymer_propagator(n_beads, omega, time_step).double()
    return torch.stack([C.t() @ torch.diag(P[:, i, j]) @ C for (i, j) in ((0, 0), (0, 1), (1, 0), (1, 1))]).float().contiguous()


class RingPolymer(VelocityVerlet):
    """md/integrators.py:113-229.  ``positions`` / ``momenta`` are [n_beads, n_atoms, 3].

    Single process: all beads local.  Bead-parallel (``group`` given, one contiguous bead chunk per rank,
    SURVEY.md section 8(e)): the rank's [n_local, n_atoms, 3] positions and momenta are packed into one
    buffer, all-gathered ONCE per step (RCCL over xGMI; 2 x 384 KB per rank at 32 k atoms) and every
    rank evaluates only its own beads of the mixed result -- no second exchange for the back-transform.
    """

    ring_polymer = True

    def __init__(self, time_step: float, n_beads: int, temperature: float, omega: Optional[float] = None,
                 group=None, compute_fn=None):
        super().__init__(time_step)
        self.n_beads = int(n_beads)
        self.omega = float(omega) if omega is not None else KB_MD * n_beads * temperature / HBAR_MD
        self.A = ring_polymer_matrices(self.n_beads, self.omega, self.time_step)
        self.group = group
        self._compute = compute_fn or _ring_polymer_hip
        self._A_dev = None

    def _bead_range(self):
        if self.group is None:
            return 0, self.n_beads, 1
        import torch.distributed as dist
        from .parallel import shard_frames
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        lo, hi = shard_frames(self.n_beads, rank, world)
        if (hi - lo) * world != self.n_beads:
            raise ValueError("bead-parallel ring polymer needs n_beads divisible by the number of ranks")
        return lo, hi, world

    def main_step(self, state):
        q, p = state.positions, state.momenta
        lo, hi, world = self._bead_range()
        n_local = hi - lo
        if q.shape[0] != n_local:
            raise ValueError("expected %d local beads, got %d" % (n_local, q.shape[0]))
        n_atoms = q.shape[1]
        if self._A_dev is None or self._A_dev.device != q.device:
            self._A_dev = self.A.to(q.device)
        if world > 1:
            import torch.distributed as dist
            local = torch.stack([q, p]).contiguous()                      # [2, n_local, n, 3]
            allb = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(allb, local.view(-1), group=self.group)
            allb = allb.view((world,) + tuple(local.shape))                # [world, 2, n_local, n, 3]
            q_all = allb[:, 0].reshape(self.n_beads, n_atoms, 3).contiguous()
            p_all = allb[:, 1].reshape(self.n_beads, n_atoms, 3).contiguous()
        else:
            q_all, p_all = q.contiguous(), p.contiguous()
        q_new, p_new = self._compute(q_all, p_all, state.masses, self._A_dev, lo, n_local)
        state.positions, state.momenta = q_new, p_new


def _ring_polymer_hip(q_all, p_all, masses, A, bead0, n_local, q_out=None, p_out=None, reference_positions=None,
                      max_displacement=0.0, flag=None):
    B, n_atoms = int(q_all.shape[0]), int(q_all.shape[1])
    m = _flat_masses(masses, n_atoms).to(q_all.device)
    if q_out is None:
        q_out = torch.empty((n_local, n_atoms, 3), dtype=torch.float32, device=q_all.device)
        p_out = torch.empty_like(q_out)
    with torch.cuda.device(q_all.device):
        check(lib().spk_md_ring_polymer_step_f32(fptr(q_all), fptr(p_all), fptr(m), fptr(A), B, n_atoms, int(bead0), int(n_local),
                                                 fptr(q_out), fptr(p_out), fptr(reference_positions), float(max_displacement) ** 2,
                                                 _lib.iptr(flag, torch.int32) if flag is not None else None, stream()))
    return q_out, p_out


def pile_matrices(n_beads: int, omega: float, time_step: float, time_constant: float, thermostat_centroid: bool = True,
                  damping_factor: float = 1.0) -> torch.Tensor:
    """M [2, B, B] = (C^T diag(c1) C, C^T diag(c2)) of the PILE-L thermostat (md/simulation_hooks/thermostats_rpmd.py:66-92):
    gamma_k = 2 omega_k (centroid: 1 / time_constant) x damping factor, c1 = exp(-dt/2 gamma), c2 = sqrt(1 - c1^2); the
    transform to normal modes, the scaling and the back-transform folded into bead-space matrices (float64 -> float32)."""
    C = normal_mode_matrix(n_beads)
    on = 2.0 * omega * torch.sin(torch.arange(n_beads).float() * math.pi / n_beads)
    gamma = 2.0 * on.double()
    if thermostat_centroid:
        gamma[0] = 1.0 / time_constant
    gamma = gamma * damping_factor
    c1 = torch.exp(-0.5 * time_step * gamma)
    c2 = torch.sqrt(1.0 - c1 ** 2)
    return torch.stack([C.t() @ torch.diag(c1) @ C, C.t() @ torch.diag(c2)]).float().contiguous()


def _pile_hip(p_all, masses, M, noise_scale, seed, step, step_dev, which, bead0, n_local, p_out=None):
    B, n_atoms = int(p_all.shape[0]), int(p_all.shape[1])
    m = _flat_masses(masses, n_atoms).to(p_all.device)
    if p_out is None:
        p_out = torch.empty((n_local, n_atoms, 3), dtype=torch.float32, device=p_all.device)
    with torch.cuda.device(p_all.device):
        check(lib().spk_md_pile_f32(fptr(p_all), fptr(m), fptr(M), float(noise_scale), int(seed), int(step),
                                    _lib.iptr(step_dev) if step_dev is not None else None, int(which), B, n_atoms, int(bead0), int(n_local),
                                    fptr(p_out), stream()))
    return p_out


class PILELocalThermostat:
    """Mirror of the reference's ``PILELocalThermostat`` (md/simulation_hooks/thermostats_rpmd.py:33-119; constructor
    arguments and the two application points of md/simulation_hooks/thermostats.py:97-123) for the device ring polymer:
    ``apply(state, step, which)`` replaces the momenta by ``C^T (c1 C p + sqrt(m kB n T) c2 xi)``.

    Bead-parallel (``group``): ONE all-gather of the momenta per application; the noise is a counter-based stream
    (Philox keyed by seed / step / atom / mode, ``spk_md_pile_f32``) that every rank regenerates identically, so there is
    no second exchange and the trajectory does not depend on the number of ranks.  ``step_dev`` (a device int64 word the
    caller increments inside its captured step) makes replays of a HIP graph draw fresh noise."""

    ring_polymer = True

    def __init__(self, temperature_bath: float, time_constant: float, thermostat_centroid: bool = True, damping_factor: float = 1.0,
                 seed: int = 0, group=None, compute_fn=None, fs: float = FS_MD, kb: float = KB_MD):
        # ``time_constant`` is in FEMTOSECONDS like the reference's (LangevinThermostat.__init__ multiplies by spk_units.fs,
        # md/simulation_hooks/thermostats.py); ``fs`` / ``kb`` are 1 fs and Boltzmann's constant in the unit system of the
        # state tensors (defaults: the reference's MD internal units kJ/mol, nm, Dalton => ps)
        self.temperature_bath, self.time_constant = float(temperature_bath), float(time_constant) * float(fs)
        self.kb = float(kb)
        self.thermostat_centroid, self.damping_factor = bool(thermostat_centroid), float(damping_factor)
        self.seed, self.group = int(seed), group
        self._compute = compute_fn or _pile_hip
        self.M = None
        self._M_dev = None

    def init(self, integrator: RingPolymer):
        """``_init_thermostat``: coefficients from the normal-mode frequencies of the integrator."""
        self.n_beads = integrator.n_beads
        self.M = pile_matrices(self.n_beads, integrator.omega, integrator.time_step, self.time_constant, self.thermostat_centroid,
                               self.damping_factor)
        self.noise_scale = math.sqrt(self.kb * self.n_beads * self.temperature_bath)
        self._range = integrator._bead_range if self.group is None else RingPolymer(integrator.time_step, self.n_beads, 1.0, omega=1.0,
                                                                                    group=self.group)._bead_range
        return self

    def apply(self, state, step: int = 0, which: int = 0, step_dev=None, out=None):
        p = state.momenta
        lo, hi, world = self._range()
        n_local = hi - lo
        if self._M_dev is None or self._M_dev.device != p.device:
            self._M_dev = self.M.to(p.device)
        if world > 1:
            import torch.distributed as dist
            local = p.contiguous()
            allb = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(allb.view(-1), local.view(-1), group=self.group)
            p_all = allb.reshape(self.n_beads, p.shape[1], 3)
        else:
            p_all = p.contiguous()
        state.momenta = self._compute(p_all, state.masses, self._M_dev, self.noise_scale, self.seed, step, step_dev, which, lo, n_local, out)
        return state.momenta


class MDState:
    """Minimal stand-in for ``schnetpack.md.System`` (md/system.py): the tensors the integrators touch."""

    def __init__(self, positions, momenta, masses, forces=None):
        self.positions, self.momenta, self.masses = positions, momenta, masses
        self.forces = forces if forces is not None else torch.zeros_like(positions)


class NVESimulation:
    """The inner loop of ``md.Simulator.simulate`` (md/simulator.py:124-157) for plain NVE dynamics of ONE
    batch of systems on one GPU, everything resident on the device.  One MD step is ONE HIP-graph replay

        kick + drift + skin test (1 kernel)  ->  force call on the current list  ->  kick (1 kernel)

    and every ``check_every`` steps (adapted to the dynamics, at most ``max_check_every``) one two-word D2H read -- the only
    host synchronisation of the loop.  Because the skin flag is read up to ``check_every`` steps AFTER it came up, the rebuild
    threshold is ``shell / 2 - margin`` with ``margin`` at least twice (usually four times) ``check_every`` x the largest
    one-step displacement seen (tracked by the kick-drift kernel): when the flag is read, the forces of all steps since it
    came up were still computed with a valid list, and the list is rebuilt (and the graph re-captured) before the next
    step.  A displacement that exceeds the margin raises.

    Batches of small isolated molecules (no cell, at most 28 atoms each -- every pair of a molecule then fits the molecule-
    resident kernels) skip the skin machinery altogether (``complete_list="auto"``): the list holds EVERY intramolecular pair,
    which is a Verlet list with an infinite skin -- pairs beyond the cutoff contribute exactly zero (the kernels do not even give
    them a tile) -- so it never has to be rebuilt, the kick-drift kernel tests nothing, and the loop is back-to-back graph
    replays without a host synchronisation.

    ``inputs`` is the batch dict on the device with ``_positions`` [N,3], ``_atomic_numbers``, ``_idx_m``,
    ``_n_atoms`` and (periodic) ``_cell`` / ``_pbc``; positions, masses, time step and the model's energy
    must share one unit system (forces = -dE/dpositions)."""

    def __init__(self, model, inputs, masses, time_step, cutoff, cutoff_shell=1.0, use_graph=True, max_check_every=4,
                 complete_list="auto"):
        from . import properties
        from .neighborlist import NeighborListMD
        self.P = properties
        self._complete = self._wants_complete_list(inputs, complete_list)
        self.max_check_every = max(int(max_check_every), 1)
        self.check_every = 1            # raised once the displacement scale is known
        self.model = model.eval()
        self.inputs = dict(inputs)
        R = inputs[properties.R].detach().float().contiguous().clone()
        self._time_step = time_step
        self._setup_state(R, masses)
        self.nl = NeighborListMD(cutoff, cutoff_shell, filter_buffer=False)
        self.use_graph = use_graph
        self.flag = torch.zeros(2, dtype=torch.int32, device=R.device)
        self.n_molecules = int(inputs[properties.n_atoms].shape[0])
        self.margin = 0.25 * cutoff_shell      # adapted to 4 x the largest one-step displacement once steps have run
        self.energy = None
        self.graph = None
        self.n_captures = 0
        self.t_rebuild = 0.0          # wall time spent in list rebuilds + graph re-captures (synchronised)
        self._lists = None
        self._rebuild()

    def _setup_state(self, R, masses):
        self.state = MDState(R.unsqueeze(0), torch.zeros_like(R).unsqueeze(0), masses.float().reshape(1, -1, 1))
        self.integrator = VelocityVerlet(self._time_step)

    # -- complete intramolecular lists ---------------------------------------------------------
    MAX_COMPLETE_ATOMS = 28        # 28 * 27 / 2 = 378 pairs <= the 384 pairs a group of the molecule-resident kernels holds

    def _wants_complete_list(self, inputs, mode) -> bool:
        P = self.P
        if mode is False or mode is None:
            return False
        pbc = inputs.get(P.pbc)
        periodic = inputs.get(P.cell) is not None and pbc is not None and bool(pbc.any())
        small = int(inputs[P.n_atoms].max()) <= self.MAX_COMPLETE_ATOMS
        if mode is True and (periodic or not small):
            raise ValueError("complete_list=True needs isolated molecules of at most %d atoms" % self.MAX_COMPLETE_ATOMS)
        return (not periodic) and small

    @staticmethod
    def complete_pair_list(idx_m: torch.Tensor, n_atoms: torch.Tensor):
        """Every ordered pair (i, j), i != j, of atoms of the same molecule; idx_i ascending, idx_j ascending within a row --
        the order of the reference's neighbour lists (atoms of a molecule are contiguous)."""
        dev = idx_m.device
        N = int(idx_m.shape[0])
        start = torch.cumsum(n_atoms, 0) - n_atoms                   # first atom of every molecule
        counts = n_atoms[idx_m] - 1                                  # partners per atom
        idx_i = torch.repeat_interleave(torch.arange(N, device=dev), counts)
        seg = torch.cumsum(counts, 0) - counts
        k = torch.arange(int(idx_i.shape[0]), device=dev) - seg[idx_i]
        first = start[idx_m[idx_i]]
        idx_j = first + k + (k >= (idx_i - first)).long()
        return idx_i, idx_j

    # -- pieces of one step ------------------------------------------------------------------
    def _flatR(self):
        return self.state.positions.view(-1, 3)

    def _call_inputs(self):
        call = dict(self.inputs)
        call.update(self._lists)
        call[self.P.R] = self._flatR()                    # the state tensor itself: no copy per step
        call["_n_molecules"] = self.n_molecules
        return call

    def _force_eval(self):
        out = self.model(self._call_inputs())
        with torch.no_grad():
            self._f.copy_(out["forces"].detach())
            self._e.copy_(out["energy"].detach())

    def _prepare_plan(self):
        """Edge plan (CSR, reverse map, skin-filter decision) of the current list without a full force call: warms the
        plan cache of the operator library outside any graph capture (one host sync per new list)."""
        P = self.P
        R = self._flatR()
        ii, jj = self._lists[P.idx_i], self._lists[P.idx_j]
        with torch.no_grad():
            r = torch.ops.spk_hip.pairwise(R.detach(), ii, jj, self._lists.get(P.offsets))
            rep = getattr(self.model, "representation", None)
            cutoff = 0.0
            if rep is not None and hasattr(rep, "cutoff_fn") and hasattr(rep.cutoff_fn, "cutoff_value"):
                cutoff = float(rep.cutoff_fn.cutoff_value())
            torch.ops.spk_hip.edge_plan(ii, jj, int(R.shape[0]), r, cutoff)
            if getattr(self.model, "_potential_forces", False):      # the energy-store decision of the fused potential: one D2H, now
                torch.ops.spk_hip.potential_plan(ii, jj, int(R.shape[0]), self.inputs[P.idx_m], self.n_molecules)

    def _step_body(self):
        if self._complete:
            self.integrator.first_half_and_main_step(self.state, True)
        else:
            thr = max(0.5 * self.nl.cutoff_shell - self.margin, 0.0)
            self.integrator.first_half_and_main_step(self.state, True, self.nl.previous_positions, thr, self.flag)
        self._force_eval()
        self.integrator.half_step(self.state)

    def _rebuild(self, new_list=True):
        import time
        t0 = time.perf_counter()
        P = self.P
        self.graph = None
        if new_list and self._complete:
            ii, jj = self.complete_pair_list(self.inputs[P.idx_m], self.inputs[P.n_atoms])
            self._lists = {P.idx_i: ii, P.idx_j: jj, P.offsets: torch.zeros(ii.shape[0], 3, device=ii.device)}
            self.nl.n_builds += 1
        elif new_list:
            self.inputs[P.R] = self._flatR()
            self.nl._list = None
            self._lists = self.nl.get_neighbors(self.inputs)
            self.flag.zero_()
        if not new_list:
            pass
        elif self.energy is None:
            out = self.model(self._call_inputs())              # first call: builds the plan, sizes the outputs
            self._f = self._force_buffer(out["forces"].detach())
            self._e = out["energy"].detach().clone()
            self.energy = self._e
        else:
            self._prepare_plan()                               # plan of the new list (host syncs) outside any capture
        if self.use_graph:
            # capture records without executing: positions / momenta are untouched
            if self.n_captures == 0:
                torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body()
            self.graph = g
            self.n_captures += 1
        torch.cuda.synchronize(self.flag.device)
        self.t_rebuild += time.perf_counter() - t0

    def _force_buffer(self, f):
        """Static force buffer the captured force call writes into ([N, 3]; the state's ``forces`` is a view of it)."""
        buf = f.clone()
        self.state.forces = buf.view(self.state.positions.shape)
        return buf

    def _one_step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_body()

    def step(self, n_steps=1):
        import time
        if self._complete:                     # nothing to watch: back-to-back replays, no host synchronisation
            for _ in range(n_steps):
                self._one_step()
            return
        done = 0
        while done < n_steps:
            k = min(self.check_every, n_steps - done)
            t0 = time.perf_counter()
            for _ in range(k):
                self._one_step()
            done += k
            moved, step_bits = self.flag.tolist()              # the one host sync of the chunk
            t_step = (time.perf_counter() - t0) / k
            step_disp = math.sqrt(struct.unpack("f", struct.pack("i", step_bits))[0])
            if k * step_disp > self.margin:
                raise RuntimeError("MD: an atom moved by up to %.3g per step over %d unchecked steps, more than the skin margin %.3g: "
                                   "reduce the time step or increase cutoff_shell" % (step_disp, k, self.margin))
            shell = self.nl.cutoff_shell
            # steps between two looks at the flag: the host round trip (tens of microseconds) matters for sub-millisecond steps
            # only, and every unchecked step costs margin (= earlier rebuilds): a few for small systems, one for large ones,
            # never more than keep 2 k x the step displacement below a twentieth of the skin
            kmax = int(0.05 * shell / (2.0 * step_disp)) if step_disp > 0.0 else self.max_check_every
            self.check_every = max(1, min(self.max_check_every, kmax, int(1.2e-3 / max(t_step, 1e-6))))
            # hysteresis: margin >= 2 x check_every x the largest one-step displacement at all times (and a floor that grows with
            # check_every: speeds may still be ramping up), re-tuned (= one re-capture, the threshold is baked into the captured
            # kernel) only when the displacement scale changed by 2x
            need = min(max(2.0 * self.check_every * step_disp, 0.0125 * self.check_every * shell), 0.45 * shell)
            retune = need > self.margin or 8.0 * need < self.margin
            if retune:
                self.margin = min(2.0 * need, 0.45 * shell)
            if moved:
                self._rebuild(True)
            elif retune:
                self._rebuild(False)

    def kinetic_energy(self):
        p, m = self.state.momenta, self.state.masses.reshape(1, -1, 1)
        return 0.5 * (p * p / m).sum()

    def total_energy(self):
        return float(self.energy.sum() + self.kinetic_energy())


class RPMDSimulation(NVESimulation):
    """Ring-polymer MD (md/integrators.py:113-229) of ``n_beads`` replicas of ONE batch of systems: the beads are folded
    into the batch dimension exactly as the reference does (md/calculators/base_calculator.py:166-183), so one force call
    and one device neighbour list serve all beads of a rank.  Single process -- one step is one graph replay:

        [PILE-L]  ->  kick (p += dt/2 F)  ->  ring-polymer main step (k_md_ring_polymer: bead mixing + skin test)  ->
        force call of all beads  ->  kick  ->  [PILE-L]

    The conserved quantity (no thermostat) is the ring-polymer Hamiltonian
    ``sum_b [p_b^2 / 2m + V(q_b)] + sum_b 1/2 m omega^2 |q_b - q_{b+1}|^2`` (``total_energy``).

    **Bead-parallel** (``group`` given; SURVEY.md section 8(e): one contiguous chunk of ``n_beads / world`` beads per rank,
    each rank with its own neighbour list and its own force-call graph; the normal-mode mixing
    md/utils/normal_model_transformation.py:70-98 is the only thing that couples beads).  Two exchange schemes:

    * ``exchange="state"`` -- every rank holds ONLY its beads.  One all-gather of the packed (positions, momenta) in the
      ring-polymer main step (``RingPolymer(group=...)``) and one all-gather of the momenta per application of the
      thermostat (``PILELocalThermostat(group=...)``): 1 + applications = 3 collectives per NVT step, 1 per NVE step --
      the reference's three exchange points (md/simulator.py:126-150), nothing more.
    * ``exchange="forces"`` -- every rank carries the integrator state of ALL beads (``n_beads x N x 3`` floats: 3 MB at
      configs[4]) and repeats the element-wise integrator / thermostat arithmetic, which is bit-identical on every rank
      (deterministic kernels, counter-based noise); only the FORCE CALL is sharded, so the one exchange of a step is the
      all-gather of the forces: 1 collective per step with or without the thermostat.

    Collectives are counted in ``n_collectives``.  The force call (+ the kernels next to it) of a rank is a HIP graph; the
    collectives run between the graph segments (RCCL on its own stream; ``gloo`` staged through the host so that two ranks
    can share one device in a test)."""

    def __init__(self, model, inputs, masses, time_step, n_beads, cutoff, temperature=300.0, omega=None,
                 cutoff_shell=1.0, use_graph=True, thermostat: Optional["PILELocalThermostat"] = None, complete_list="auto",
                 group=None, exchange: str = "state"):
        from . import properties as P
        if exchange not in ("state", "forces"):
            raise ValueError("exchange must be 'state' or 'forces'")
        self.thermostat = thermostat
        self.n_beads = int(n_beads)
        self.group, self.exchange = group, exchange
        self.n_collectives = 0
        self._rp = RingPolymer(time_step, self.n_beads, temperature, omega=omega, group=group)
        self._lo, hi, self._world = self._rp._bead_range()
        # the distributed code path: several ranks -- or ONE rank with a group when SPK_MD_FORCE_COLLECTIVES=1 (the RCCL smoke test of a one-GPU
        # box: every all-gather then executes, as an identity, on the real back-end)
        self._dist = self._world > 1 or (group is not None and os.environ.get("SPK_MD_FORCE_COLLECTIVES") == "1")
        self.n_local = B = hi - self._lo                      # beads in THIS rank's batch
        N = int(inputs[P.R].shape[0])
        n_mol = int(inputs[P.n_atoms].shape[0])
        rep = dict(inputs)
        rep[P.R] = inputs[P.R].detach().float().repeat(B, 1)
        rep[P.Z] = inputs[P.Z].repeat(B)
        rep[P.idx_m] = (inputs[P.idx_m][None, :] + n_mol * torch.arange(B, device=inputs[P.idx_m].device)[:, None]).reshape(-1)
        rep[P.n_atoms] = inputs[P.n_atoms].repeat(B)
        if inputs.get(P.cell) is not None:
            rep[P.cell] = inputs[P.cell].reshape(-1, 3, 3).repeat(B, 1, 1)
        if inputs.get(P.pbc) is not None:
            rep[P.pbc] = inputs[P.pbc].reshape(-1, 3).repeat(B, 1).reshape(-1)
        self._n1 = N
        super().__init__(model, rep, masses, time_step, cutoff, cutoff_shell, use_graph, complete_list=complete_list)

    # -- state ---------------------------------------------------------------------------------
    @property
    def _replicated(self) -> bool:
        return self._dist and self.exchange == "forces"

    def _setup_state(self, R, masses):
        Bl, N, dev = self.n_local, self._n1, R.device
        m = masses.float().reshape(1, -1, 1)
        if self._replicated:
            # integrator state of ALL beads on every rank (they start from the same geometry); the force call reads the
            # rank's rows of it in place
            Ball = self.n_beads
            self._q_all = R.view(Bl, N, 3)[:1].repeat(Ball, 1, 1).contiguous()
            self._p_all = torch.zeros(Ball, N, 3, device=dev)
            self._f_all = torch.zeros(Ball, N, 3, device=dev)
            self.full = MDState(self._q_all, self._p_all, m, self._f_all)
            lo = self._lo
            self.state = MDState(self._q_all[lo:lo + Bl], self._p_all[lo:lo + Bl], m, self._f_all[lo:lo + Bl])
            nb = Ball
        else:
            self.state = MDState(R.view(Bl, N, 3), torch.zeros(Bl, N, 3, device=dev), m)
            nb = Bl
        self.integrator = self._rp
        self._m_rep = m.reshape(-1).repeat(Bl).contiguous()
        self._qt = torch.empty(nb, N, 3, device=dev)
        self._pt = torch.empty(nb, N, 3, device=dev)
        self._A = self._rp.A.to(dev)
        if self._dist and not self._replicated:
            self._pack = torch.empty(2, Bl, N, 3, device=dev)                    # (q, p) of the rank, one message
            self._gath = torch.empty(self._world, 2, Bl, N, 3, device=dev)
            self._pgath = torch.empty(self._world, Bl, N, 3, device=dev)
        if self.thermostat is not None:          # NVT: PILE-L at step begin and end (md/simulator.py:126-150)
            self.thermostat.init(self._rp)
            self._M = self.thermostat.M.to(dev)
            self._stepc = torch.zeros(1, dtype=torch.int64, device=dev)      # step counter on the device: fresh noise per graph replay

    def _force_buffer(self, f):
        if not self._replicated:
            return super()._force_buffer(f)
        buf = self.state.forces.view(-1, 3)        # the rank's rows of the all-bead force tensor, written in place
        buf.copy_(f)
        return buf

    # -- the one collective primitive ------------------------------------------------------------
    def _all_gather(self, out: torch.Tensor, local: torch.Tensor):
        """``out[r] = local of rank r`` (contiguous buffers).  RCCL directly on the device buffers; other back-ends (gloo in
        the tests: it has no device all-gather) through pinned host staging."""
        import torch.distributed as dist
        self.n_collectives += 1
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=self.group)
            return
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out.view(-1), local.detach().cpu().view(-1), group=self.group)
        out.copy_(h_out)

    # -- pieces of a step --------------------------------------------------------------------------
    def _thermostat(self, which):
        th = self.thermostat
        if self._replicated:                       # all beads, every rank, same counter-based noise: no exchange
            st = self.full
            _pile_hip(st.momenta, st.masses, self._M, th.noise_scale, th.seed, 0, self._stepc, which, 0, self.n_beads, self._pt)
        elif self._dist:                      # the rank's beads from everybody's momenta: ONE all-gather
            st = self.state
            self._all_gather(self._pgath, st.momenta)
            _pile_hip(self._pgath.view(self.n_beads, self._n1, 3), st.masses, self._M, th.noise_scale, th.seed, 0, self._stepc, which,
                      self._lo, self.n_local, self._pt)
        else:
            st = self.state
            _pile_hip(st.momenta, st.masses, self._M, th.noise_scale, th.seed, 0, self._stepc, which, 0, self.n_beads, self._pt)
        with torch.no_grad():
            st.momenta.copy_(self._pt)

    def _skin(self):
        thr = max(0.5 * self.nl.cutoff_shell - self.margin, 0.0)
        ref = None if self._complete else self.nl.previous_positions       # complete lists: nothing to watch
        return ref, thr, (None if self._complete else self.flag)

    def _mix(self):
        """kick + ring-polymer main step of the beads this rank integrates (skin test on the beads of its list)."""
        ref, thr, flag = self._skin()
        if self._replicated:
            st = self.full
            self.integrator.half_step(st)
            _ring_polymer_hip(st.positions, st.momenta, st.masses, self._A, 0, self.n_beads, self._qt, self._pt)
            with torch.no_grad():
                st.positions.copy_(self._qt)
                st.momenta.copy_(self._pt)
            if ref is not None:                    # skin criterion of the rank's own beads (a drift of zero length: test only)
                loc = self.state
                with torch.cuda.device(loc.positions.device):
                    check(lib().spk_md_kick_drift_f32(fptr(loc.positions), fptr(loc.momenta), None, fptr(self._m_rep), 0.0,
                                                      loc.positions.numel() // 3, fptr(ref), float(thr) ** 2,
                                                      _lib.iptr(flag, torch.int32), stream()))
            return
        st = self.state
        self.integrator.half_step(st)
        if self._dist:
            with torch.no_grad():
                self._pack[0].copy_(st.positions)
                self._pack[1].copy_(st.momenta)
            self._all_gather(self._gath, self._pack)
            q_all = self._gath[:, 0].reshape(self.n_beads, self._n1, 3)          # [world, n_local] -> beads (copy: strided)
            p_all = self._gath[:, 1].reshape(self.n_beads, self._n1, 3)
        else:
            q_all, p_all = st.positions, st.momenta
        _ring_polymer_hip(q_all, p_all, st.masses, self._A, self._lo, self.n_local, self._qt, self._pt, ref, thr, flag)
        with torch.no_grad():
            st.positions.copy_(self._qt)
            st.momenta.copy_(self._pt)

    def _finish(self):
        """forces of the new positions -> second kick (-> thermostat, step counter)."""
        if self._replicated:
            self._all_gather(self._f_all.view(self._world, -1), self.state.forces)
            self.integrator.half_step(self.full)
        else:
            self.integrator.half_step(self.state)
        if self.thermostat is not None:
            self._thermostat(1)
            with torch.no_grad():
                self._stepc.add_(1)

    def _step_body(self):
        """single process: the whole step (one graph).  Bead-parallel: only the force evaluation (the graph segment between
        the collectives); ``_one_step`` adds the rest."""
        if self._dist:
            self._force_eval()
            return
        if self.thermostat is not None:
            self._thermostat(0)
        self._mix()
        self._force_eval()
        self._finish()

    def _one_step(self):
        if not self._dist:
            return super()._one_step()
        if self.thermostat is not None:
            self._thermostat(0)
        self._mix()
        super()._one_step()                        # force call of the rank's beads: graph replay
        self._finish()

    @property
    def collectives_per_step(self) -> int:
        if not self._dist:
            return 0
        if self._replicated:
            return 1
        return 1 + (2 if self.thermostat is not None else 0)

    def spring_energy(self):
        """Spring energy of the beads this process holds (all of them unless ``exchange="state"`` over several ranks, where
        the links to the neighbouring ranks' beads are not visible locally)."""
        st = self.full if self._replicated else self.state
        q, m = st.positions, st.masses.reshape(1, -1, 1)
        d = q - torch.roll(q, -1, 0)
        return 0.5 * self._rp.omega ** 2 * (m * d * d).sum()

    def total_energy(self):
        return float(self.energy.sum() + self.kinetic_energy() + self.spring_energy())
