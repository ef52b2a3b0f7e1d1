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


# ----------------------------------------------------------------------------- PILE-L thermostat
def pile_coefficients(n_beads: int, omega: float, dt: float, time_constant: float, thermostat_centroid: bool = True,
                      damping_factor: float = 1.0):
    """(c1 [B], c2 [B]) of md/simulation_hooks/thermostats_rpmd.py:66-92: gamma_k = 2 omega_k (centroid: 1 / tau),
    times the TRPMD damping factor; c1 = exp(-dt/2 gamma), c2 = sqrt(1 - c1^2)."""
    omega_normal, _ = ring_polymer_propagator(n_beads, omega, dt)
    gamma = 2.0 * omega_normal.double()
    if thermostat_centroid:
        gamma[0] = 1.0 / time_constant
    gamma = gamma * damping_factor
    c1 = torch.exp(-0.5 * dt * gamma)
    return c1, torch.sqrt(1.0 - c1 ** 2)


def pile_apply(p: Tensor, masses: Tensor, C: Tensor, c1: Tensor, c2: Tensor, kB_nT: float, noise_nm: Tensor) -> Tensor:
    """thermostats_rpmd.py:102-119 with the normal-mode noise given explicitly: to normal modes, c1 p + sqrt(m kB n T) c2 xi,
    back.  p [B, n, 3], noise_nm [B, n, 3] (standard normals per mode), masses broadcastable [1, n, 1]."""
    B = p.shape[0]
    C = C.to(p.dtype)
    pn = (C @ p.reshape(B, -1)).view(p.shape)
    pn = c1.to(p.dtype)[:, None, None] * pn + torch.sqrt(masses * kB_nT) * c2.to(p.dtype)[:, None, None] * noise_nm
    return (C.t() @ pn.reshape(B, -1)).view(p.shape)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11) on numpy uint32 arrays / scalars: the counter-based generator of
    spk_md_pile_f32 restated (no reference counterpart -- the reference draws torch.randn_like)."""
    import numpy as np
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & np.uint64(0xFFFFFFFF), np.uint64(k1) & np.uint64(0xFFFFFFFF)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def pile_noise(n_beads: int, n_atoms: int, seed: int, step: int, which: int) -> Tensor:
    """The normal-mode noise xi [B, n, 3] (float64) that spk_md_pile_f32 generates: counter = (component index t = 3 atom + c,
    (t >> 32) ^ (mode pair << 8) ^ which, step lo, step hi), key = seed; Box-Muller on the first two words gives the modes
    2 k2 (cos) and 2 k2 + 1 (sin)."""
    import numpy as np
    t = np.arange(3 * n_atoms, dtype=np.uint64)
    xi = np.zeros((n_beads, 3 * n_atoms))
    for k2 in range((n_beads + 1) // 2):
        w0, w1, _, _ = philox4x32_10(t, (t >> np.uint64(32)) ^ np.uint64((k2 << 8) ^ which), np.uint64(step & 0xFFFFFFFF), np.uint64(step >> 32),
                                     seed & 0xFFFFFFFF, seed >> 32)
        u = ((w0 >> np.uint64(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
        v = (w1 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        r = np.sqrt(-2.0 * np.log(u.astype(np.float64)))
        xi[2 * k2] = r * np.cos(2.0 * np.pi * v.astype(np.float64))
        if 2 * k2 + 1 < n_beads:
            xi[2 * k2 + 1] = r * np.sin(2.0 * np.pi * v.astype(np.float64))
    return torch.from_numpy(xi).view(n_beads, n_atoms, 3)
