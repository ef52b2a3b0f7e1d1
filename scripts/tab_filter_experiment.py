"""EXPERIMENT: tabulated SchNet filters -- time and error of the table-driven continuous-filter
convolution (csrc/spk_tabfilter.hip) beside the fp32-MFMA contract path, at cfg 2 (256 aspirin frames) and on the 31 944-atom
water box.  Eval-only, default off; writes one JSON object (profiles/r03_tabulated_filter_experiment.json).

Table: W_l(d) f_c(d) per channel = (ssp(phi(d) W1^T + b1) W2^T + b2) f_c(d) (schnet.py:60-62) and its slope, evaluated on the HOST
in float64 from the model's weights at n_knots equidistant points of [0, cutoff] (the filter is identically zero beyond).
"""
import ctypes, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from schnetpack_amd import _lib, model as M, ops, synthetic as S, tabulate

dev = torch.device("cuda:0")
L = _lib.lib()
F, n_rbf, cutoff = 128, 20, 5.0


def filter_fp64(inter, rb, d):
    """(W f_c, d(W f_c)/dd) [len(d), F] in float64 through autograd of the exact formula."""
    d = d.double().clone().requires_grad_(True)
    off, wid = rb.offsets.double().cpu(), rb.widths.double().cpu()
    phi = torch.exp(-0.5 / wid ** 2 * (d[:, None] - off[None, :]) ** 2)
    w1, b1 = inter.filter_network[0].weight.double().cpu(), inter.filter_network[0].bias.double().cpu()
    w2, b2 = inter.filter_network[1].weight.double().cpu(), inter.filter_network[1].bias.double().cpu()
    hdn = torch.nn.functional.softplus(phi @ w1.t() + b1) - math.log(2.0)
    fc = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0) * (d < cutoff)
    W = (hdn @ w2.t() + b2) * fc[:, None]
    dW = torch.stack([torch.autograd.grad(W[:, c].sum(), d, retain_graph=True)[0] for c in range(F)], 1)
    return W.detach(), dW.detach()


def build_table(inter, rb, n_knots):
    d = torch.linspace(0.0, cutoff, n_knots, dtype=torch.float64)
    W, dW = filter_fp64(inter, rb, d)
    step = cutoff / (n_knots - 1)
    return tabulate.pack_knots(W, dW * step).float().contiguous()          # [n_knots, F, 4] = (v_n, m_n, v_n+1 - v_n, m_n+1)


def hermite(table, d, step):
    """The kernel's interpolation restated in float64 (value and slope) -- the error of the TABLE itself."""
    u = d.double() / step
    n = u.floor().clamp(max=table.shape[0] - 2).long()
    s = (u - n)[:, None]
    k = table[n].double()
    h10, h01, h11 = s ** 3 - 2 * s ** 2 + s, -2 * s ** 3 + 3 * s ** 2, s ** 3 - s ** 2
    W = k[..., 0] + h10 * k[..., 1] + h01 * k[..., 2] + h11 * k[..., 3]
    dh10, dh01, dh11 = 3 * s ** 2 - 4 * s + 1, -6 * s ** 2 + 6 * s, 3 * s ** 2 - 2 * s
    dW = (dh10 * k[..., 1] + dh01 * k[..., 2] + dh11 * k[..., 3]) / step
    return W, dW


def event_time_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


torch.manual_seed(0)
model = M.build_model("schnet", F, 3, n_rbf, cutoff).eval()
inter, rb = model.representation.interactions[0], model.representation.radial_basis
res = {"what": "cubic-Hermite table of W_l(d) f_c(d) (value + slope per knot and channel, built in float64 on the host) in place of the filter network; "
               "forward of ONE interaction's cfconv y = sum_j h_j * W(d_ij); eval-only experiment, default off", "table_error": [], "workloads": []}
g = torch.Generator().manual_seed(1)
dq = torch.rand(20000, generator=g, dtype=torch.float64) * cutoff
Wx, dWx = filter_fp64(inter, rb, dq)
for nk in (128, 256, 512, 1024):
    tab = build_table(inter, rb, nk)
    Wt, dWt = hermite(tab, dq, cutoff / (nk - 1))
    res["table_error"].append({"n_knots": nk, "table_KB_per_interaction": round(tab.numel() * 4 / 1024, 1),
                               "max_abs_err_W_over_max_W": float((Wt - Wx).abs().max() / Wx.abs().max()),
                               "max_abs_err_dW_over_max_dW": float((dWt - dWx).abs().max() / dWx.abs().max())})
model = model.to(dev)
for wl in ("aspirin256", "water31944"):
    b = S.molecule_batch("aspirin", 256, seed=0) if wl == "aspirin256" else S.water_box(n_side=22, seed=0)
    N, E = int(b["Z"].shape[0]), int(b["idx_i"].shape[0])
    ii, jj = b["idx_i"].to(dev), b["idx_j"].to(dev)
    r = (b["R"][b["idx_j"]] - b["R"][b["idx_i"]] + b["offsets"]).float().to(dev)
    h = torch.randn(N, F, device=dev)
    plan = ops.EdgePlan(ii, jj, N, r)
    y = torch.empty(N, F, device=dev)
    row = {"workload": wl, "N": N, "E": E, "tables": []}
    # exact result in float64 on the host (a sample of atoms for the water box)
    d = r.double().norm(dim=1).cpu()
    sel_atoms = torch.arange(N) if N <= 6000 else torch.randperm(N, generator=g)[:2000]
    mask = torch.isin(b["idx_i"], sel_atoms)
    We, _ = filter_fp64(inter, rb, d[mask])
    y_ref = torch.zeros(N, F, dtype=torch.float64).index_add_(0, b["idx_i"][mask], We * h.double().cpu()[b["idx_j"][mask]])[sel_atoms]
    for nk in (256, 512, 1024):
        tab = build_table(inter, rb, nk).to(dev)

        def run():
            _lib.check(L.spk_cfconv_tab_f32(plan.graph(), _lib.fptr(r), _lib.fptr(h), _lib.fptr(tab), nk, cutoff, cutoff, F, _lib.fptr(y), _lib.stream()))
        us = event_time_us(run)
        err = float((y.double().cpu()[sel_atoms] - y_ref).abs().max() / y_ref.abs().max())
        row["tables"].append({"n_knots": nk, "us_per_launch": round(us, 2), "M_edge_messages_per_s_this_kernel": round(E / us, 1),
                              "bytes_per_edge_gathered": 2560, "GB_per_s_gathered": round(E * 2560.0 / us / 1e3, 1),
                              "rel_err_y_vs_fp64_exact": err})
    # the contract path on the same inputs: the fused MFMA cfconv forward of the product (one interaction, filters recomputed in-kernel)
    lw = model.representation.interactions[0]
    rbs = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, n_rbf, rb.offsets, rb.widths, cutoff)
    _lib.profile_enable(True); _lib.profile_report()
    inp = M.batch_to_inputs(b, dev)
    for _ in range(5):
        out = model(dict(inp))
    prof = _lib.profile_report(); _lib.profile_enable(False)
    row["contract_path_kernels_us"] = {k: round(1e3 * v[1] / max(v[0], 1), 1) for k, v in prof.items() if "cfconv" in k or "schnet_mol" in k}
    # the whole eval force call (PairwiseDistances -> SchNet -> Atomwise -> Forces), contract path vs tables attached (512 knots)
    from schnetpack_amd import tabulate

    def force_call():
        o = model(dict(inp))
        return o["forces"].detach()
    f_contract = force_call().clone()
    row["force_call_contract_ms"] = round(event_time_us(force_call, reps=10) / 1e3, 4)
    tabulate.tabulate_filters(model.representation, 512)
    _lib.profile_enable(True); _lib.profile_report()
    f_tab = force_call().clone()
    prof = _lib.profile_report(); _lib.profile_enable(False)
    row["force_call_tabulated_ms"] = round(event_time_us(force_call, reps=10) / 1e3, 4)
    row["tabulated_kernels_us"] = {k: round(1e3 * v[1] / max(v[0], 1), 1) for k, v in prof.items() if "cfconv" in k}
    row["forces_tabulated_vs_contract_rel"] = float((f_tab - f_contract).abs().max() / f_contract.abs().max())
    tabulate.clear_filter_tables()
    res["workloads"].append(row)
res["conclusion"] = "see DESIGN.md section 7 (round 3): time and error of the table kernel beside the fp32-MFMA kernels of the same launch"
print(json.dumps(res))
