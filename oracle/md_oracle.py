"""CPU restatement of the MD steps on the path's edge.  TEST
INFRASTRUCTURE ONLY.  Plain torch (fp64 in the tests), each function citing the reference lines.
Pinned by tests/golden/md_ring_polymer.npz, which oracle/make_golden.py produces by executing the
reference's own ``RingPolymer._init_propagator`` / ``_main_step`` and ``NormalModeTransformer``.
"""
import math

import torch

Tensor = torch.Tensor


def half_step(p: Tensor, F: Tensor, dt: float) -> Tensor:
    """p + dt/2 F   (md/integrators.py:59-70)."""
    return p + 0.5 * F * dt


def verlet_main_step(R: Tensor, p: Tensor, masses: Tensor, dt: float) -> Tensor:
    """R + dt p / m   (md/integrators.py:97-110)."""
    return R + dt * p / masses


def normal_mode_matrix(n_beads: int) -> Tensor:
    """C[k, n] of the bead -> normal-mode transformation (md/utils/normal_model_transformation.py:38-68):
    row 0 constant, rows 1..B/2 cosines, the rest sines, the Nyquist row (-1)^n for even B, all / sqrt(B)."""
    B = n_beads
    n = torch.arange(1, B + 1, dtype=torch.float64)
    C = torch.zeros(B, B, dtype=torch.float64)
    C[0] = 1.0
    for k in range(1, B // 2 + 1):
        C[k] = math.sqrt(2.0) * torch.cos(2 * math.pi * k * n / B)
    for k in range(B // 2 + 1, B):
        C[k] = math.sqrt(2.0) * torch.sin(2 * math.pi * k * n / B)
    if B % 2 == 0:
        C[B // 2] = (-1.0) ** n
    return C / math.sqrt(B)


def ring_polymer_propagator(n_beads: int, omega: float, dt: float):
    """(omega_normal [B], propagator [B,2,2]) of md/integrators.py:152-199; fp32 sin/cos like the
    reference (``torch.arange(n).float()``)."""
    omega_normal = 2.0 * omega * torch.sin(torch.arange(n_beads).float() * math.pi / n_beads)
    odt = omega_normal * dt
    c, s = torch.cos(odt), torch.sin(odt)
    P = torch.zeros(n_beads, 2, 2)
    P[:, 0, 0] = c
    P[:, 1, 1] = c
    P[:, 0, 1] = -s * omega_normal
    P[1:, 1, 0] = s[1:] / omega_normal[1:]
    P[0, 1, 0] = dt
    return omega_normal, P


def ring_polymer_main_step(q: Tensor, p: Tensor, masses: Tensor, C: Tensor, P: Tensor):
    """md/integrators.py:204-229: to normal modes (C x), 2x2 propagation per mode (momenta and
    positions*mass mixed), back (C^T x).  q, p [B, n, 3]; masses broadcastable [1, n, 1]."""
    B = q.shape[0]
    C = C.to(q.dtype)
    P = P.to(q.dtype)
    qn = (C @ q.reshape(B, -1)).view(q.shape)
    pn = (C @ p.reshape(B, -1)).view(p.shape)
    pe = P[:, :, :, None, None]
    pn2 = pe[:, 0, 0] * pn + pe[:, 0, 1] * qn * masses
    qn2 = pe[:, 1, 0] * pn / masses + pe[:, 1, 1] * qn
    p2 = (C.t() @ pn2.reshape(B, -1)).view(p.shape)
    q2 = (C.t() @ qn2.reshape(B, -1)).view(q.shape)
    return q2, p2
