"""Pin the forward-over-reverse restatement (oracle/fm_oracle.py) against autograd's double backward through the oracle of the
hot path (oracle/spk_oracle.py, itself pinned to the live reference): the force-matching weight gradients that the reference
obtains with ``create_graph=True`` (atomistic/response.py:59-68) equal one dual-number forward + one reverse pass.  float64, CPU."""
import pytest
import torch

from oracle import fm_oracle as FM
from oracle import spk_oracle as O
from schnetpack_amd import synthetic


def _autograd_reference(kind, rep_p, head_p, b, L, Et, Ft, wE, wF, shared=False):
    fixed = ("radial_basis", "cutoff_fn")
    rp = {k: (v.double().clone().requires_grad_(not k.startswith(fixed)) if v.is_floating_point() else v) for k, v in rep_p.items()}
    hp = {k: v.double().clone().requires_grad_(True) for k, v in head_p.items()}
    R = b["R"].double().clone().requires_grad_(True)
    r_ij = O.pairwise_vectors(R, b["idx_i"], b["idx_j"], b["offsets"].double())
    if kind == "schnet":
        x = O.schnet_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, L)
    else:
        x, _ = O.painn_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, L, shared)
    E = O.atomwise_energy(x, b["idx_m"], int(b["n_mol"]), hp)
    (dEdR,) = torch.autograd.grad([E.sum()], [R], create_graph=True)
    F = -dEdR
    loss = wE * ((E - Et) ** 2).mean() + wF * ((F - Ft) ** 2).mean()
    names = [k for k, v in rp.items() if torch.is_tensor(v) and v.requires_grad]
    hnames = list(hp)
    gs = torch.autograd.grad(loss, [rp[k] for k in names] + [hp[k] for k in hnames], allow_unused=True)
    out = dict(zip(names + hnames, gs))
    return E.detach(), F.detach(), out


@pytest.mark.parametrize("kind,radial,shared", [("schnet", "gaussian", False), ("schnet", "bessel", False), ("painn", "gaussian", False),
                                                ("painn", "bessel", False), ("painn", "gaussian", True)])
def test_forward_over_reverse_equals_double_backward(kind, radial, shared):
    torch.manual_seed(3)
    F_, L, n_rbf = 16, 2, 8
    b = synthetic.molecule_batch("aspirin", n_frames=2, cutoff=5.0, seed=5)
    if kind == "schnet":
        rep_p = O.init_schnet_params(F_, L, n_rbf, 5.0, radial=radial, seed=0)
    else:
        rep_p = O.init_painn_params(F_, L, n_rbf, 5.0, radial=radial, seed=0, shared_filters=shared)
    head_p = O.init_atomwise_params(F_, seed=1)
    for k in list(rep_p):                                   # non-zero biases so that every bias gradient is exercised
        if k.endswith("bias"):
            rep_p[k] = 0.1 * torch.randn_like(rep_p[k])
    for k in list(head_p):
        if k.endswith("bias"):
            head_p[k] = 0.1 * torch.randn_like(head_p[k])
    M, N = int(b["n_mol"]), b["Z"].shape[0]
    Et, Ft = torch.randn(M, dtype=torch.float64), torch.randn(N, 3, dtype=torch.float64)
    wE, wF = 0.01, 0.99
    E_ref, F_ref, g_ref = _autograd_reference(kind, rep_p, head_p, b, L, Et, Ft, wE, wF, shared)

    if kind == "schnet":
        E, Fo, saved = FM.schnet_forward(rep_p, head_p, b, L)
    else:
        E, Fo, saved = FM.painn_forward(rep_p, head_p, b, L, shared_filters=shared)
    assert torch.allclose(E, E_ref, rtol=1e-11, atol=1e-11)
    assert torch.allclose(Fo, F_ref, rtol=1e-10, atol=1e-11)
    gE = 2 * wE * (E - Et) / M
    gF = 2 * wF * (Fo - Ft) / (3 * N)
    grads = FM.schnet_backward(saved, gE, gF) if kind == "schnet" else FM.painn_backward(saved, gE, gF)
    checked = 0
    for k, ref in g_ref.items():
        if ref is None:
            continue
        got = grads[k]
        scale = float(ref.abs().max()) + 1e-300
        err = float((got.reshape(ref.shape) - ref).abs().max()) / scale
        assert err < 1e-9, (k, err)
        checked += 1
    assert checked >= (9 * L + 5 if kind == "schnet" else 9 * L + 7)
