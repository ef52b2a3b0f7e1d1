"""Force-matching gradients by FORWARD-OVER-REVERSE, written out by hand.  TEST INFRASTRUCTURE ONLY.

The reference trains on forces by differentiating ``-dE/dR`` a second time (``atomistic/response.py:59-68``:
``create_graph=self.training``; ``task.py:166-185``): reverse over reverse through the autograd graph.  The gradient of a
loss ``L(E, F)`` w.r.t. the weights is, with ``gE = dL/dE`` and ``gF = dL/dF`` held fixed,

    dL/dtheta = d/dtheta [ sum_m gE_m E_m  +  D_t E_tot ],      t = -gF,

where ``D_t E_tot = sum_i t_i . dE_tot/dR_i`` is the directional derivative of the total energy along ``t``.  So one
*dual-number* forward pass (values and tangents along ``t``) followed by ONE ordinary reverse pass gives every weight
gradient -- no second-order autograd graph.  This file states that computation explicitly (no autograd anywhere) in the
four passes the HIP engine runs (``schnetpack_amd/csrc/spk_fm_engine.h``):

    A  values, everything later passes need is kept            (representation/schnet.py:147-173, painn.py:207-256)
    B  reverse w.r.t. the positions  -> forces                 (atomistic/response.py:59-76)
    C  tangents along t = -gF
    D  reverse of the dual graph     -> weight gradients

It is pinned by ``tests/test_fm_oracle.py`` against autograd's double backward through ``oracle/spk_oracle.py`` (which is
itself pinned against the live reference) in float64.  Key names of the returned gradients are the reference's
``state_dict`` keys.  Notation: a trailing ``1`` marks a derivative with respect to the pair distance d, a ``t`` a tangent,
``g*`` a cotangent of a value, ``h*`` a cotangent of a tangent.
"""
import math
from typing import Dict

import torch

from . import spk_oracle as O

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------ scalar functions
def act_order(name: str, k: int, z: Tensor) -> Tensor:
    """k-th derivative of shifted softplus / SiLU (nn/activations.py:9-22, F.silu)."""
    s = torch.sigmoid(z)
    if name == "ssp":
        return [torch.nn.functional.softplus(z) - math.log(2.0), s, s * (1 - s)][k]
    return [z * s, s * (1 + z * (1 - s)), s * (1 - s) * (2 + z * (1 - 2 * s))][k]


def radial_with_derivative(d: Tensor, p: Dict[str, Tensor]):
    """phi_k(d), phi_k'(d)  (nn/radial.py:11-15 Gaussian, :105-110 Bessel)."""
    if "radial_basis.freqs" in p:
        f = p["radial_basis.freqs"].to(d.dtype)
        x = d[:, None] * f
        return torch.sin(x) / d[:, None], (f * torch.cos(x) - torch.sin(x) / d[:, None]) / d[:, None]
    mu, w = p["radial_basis.offsets"].to(d.dtype), p["radial_basis.widths"].to(d.dtype)
    c = -0.5 / (w * w)
    t = d[:, None] - mu
    phi = torch.exp(c * t * t)
    return phi, 2 * c * t * phi


def cutoff_with_derivative(d: Tensor, rc: float):
    """f_c(d), f_c'(d)  (nn/cutoff.py:14-33)."""
    inside = (d < rc).to(d.dtype)
    a = math.pi / rc
    return 0.5 * (torch.cos(a * d) + 1) * inside, -0.5 * a * torch.sin(a * d) * inside


def seg_sum(x: Tensor, idx: Tensor, n: int) -> Tensor:
    return torch.zeros((n,) + tuple(x.shape[1:]), dtype=x.dtype).index_add(0, idx, x)


def geometry(batch, p, dtype):
    R = batch["R"].to(dtype)
    ii, jj = batch["idx_i"], batch["idx_j"]
    r = R[jj] - R[ii] + batch["offsets"].to(dtype)
    d = torch.sqrt((r * r).sum(1))
    u = r / d[:, None]
    phi, phi1 = radial_with_derivative(d, p)
    fc, fc1 = cutoff_with_derivative(d, float(p["cutoff_fn.cutoff"]))
    return dict(r=r, d=d, u=u, phi=phi, phi1=phi1, fc=fc, fc1=fc1, ii=ii, jj=jj, N=R.shape[0])


def tangent_geometry(g, t):
    """rt = t_j - t_i, dt = u . rt, ut = (rt - u dt) / d."""
    rt = t[g["jj"]] - t[g["ii"]]
    dt = (g["u"] * rt).sum(1)
    return dt, (rt - g["u"] * dt[:, None]) / g["d"][:, None]


def forces_from_edge_gradient(g, gd, gu=None):
    """gr_e = gd_e u_e + (gu_e - (gu_e . u_e) u_e) / d_e ;  dE/dR_a = sum_{e: j(e)=a} gr_e - sum_{e: i(e)=a} gr_e ;  F = -dE/dR."""
    gr = gd[:, None] * g["u"]
    if gu is not None:
        gr = gr + (gu - (gu * g["u"]).sum(1, keepdim=True) * g["u"]) / g["d"][:, None]
    return -(seg_sum(gr, g["jj"], g["N"]) - seg_sum(gr, g["ii"], g["N"]))


# ------------------------------------------------------------------------------------------------ energy head (atomistic/atomwise.py:69-88)
def head_forward(x, idx_m, n_mol, hp):
    pre = x @ hp["outnet.0.weight"].t() + hp["outnet.0.bias"]
    th = act_order("silu", 0, pre)
    e = th @ hp["outnet.1.weight"][0] + hp["outnet.1.bias"][0]
    return seg_sum(e, idx_m, n_mol), dict(pre=pre, th=th)


def head_backward_R(hs, hp):
    """dE_tot/dx  (every molecule's energy enters the force with weight 1)."""
    return (hp["outnet.1.weight"][0] * act_order("silu", 1, hs["pre"])) @ hp["outnet.0.weight"]


def head_dual_backward(x, xt, hs, hp, gE_atom, grads):
    """S = sum_i gE_i e_i + sum_i et_i.  Returns (gx, hx)."""
    w1, w2 = hp["outnet.0.weight"], hp["outnet.1.weight"][0]
    pre, th = hs["pre"], hs["th"]
    pret = xt @ w1.t()
    tht = act_order("silu", 1, pre) * pret
    gt = gE_atom[:, None] * w2
    ht = w2.expand_as(gt)
    grads["outnet.1.weight"] = ((gE_atom[:, None] * th).sum(0) + tht.sum(0))[None, :]
    grads["outnet.1.bias"] = gE_atom.sum()[None]
    gpre = gt * act_order("silu", 1, pre) + ht * act_order("silu", 2, pre) * pret
    hpre = ht * act_order("silu", 1, pre)
    grads["outnet.0.weight"] = gpre.t() @ x + hpre.t() @ xt
    grads["outnet.0.bias"] = gpre.sum(0)
    return gpre @ w1, hpre @ w1


# ================================================================================================ SchNet
def schnet_forward(rep_p, head_p, batch, L: int, dtype=torch.float64):
    """Passes A + B: (E [M], F [N,3], saved)."""
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in rep_p.items()}
    hp = {k: v.to(dtype) for k, v in head_p.items()}
    g = geometry(batch, p, dtype)
    ii, jj, N = g["ii"], g["jj"], g["N"]
    x = p["embedding.weight"][batch["Z"]]
    lay = []
    for l in range(L):                                                      # ---- pass A
        pre = "interactions.%d." % l
        w1, b1 = p[pre + "filter_network.0.weight"], p[pre + "filter_network.0.bias"]
        w2, b2 = p[pre + "filter_network.1.weight"], p[pre + "filter_network.1.bias"]
        a, a1 = g["phi"] @ w1.t() + b1, g["phi1"] @ w1.t()                   # value and d-derivative of the filter network
        z, z1 = act_order("ssp", 0, a), act_order("ssp", 1, a) * a1
        gf, gf1 = z @ w2.t() + b2, z1 @ w2.t()
        Wf, Wf1 = gf * g["fc"][:, None], gf1 * g["fc"][:, None] + gf * g["fc1"][:, None]
        h = x @ p[pre + "in2f.weight"].t()
        y = seg_sum(h[jj] * Wf, ii, N)
        p3 = y @ p[pre + "f2out.0.weight"].t() + p[pre + "f2out.0.bias"]
        s = act_order("ssp", 0, p3)
        v = s @ p[pre + "f2out.1.weight"].t() + p[pre + "f2out.1.bias"]
        lay.append(dict(x=x, h=h, a=a, a1=a1, z=z, z1=z1, Wf=Wf, Wf1=Wf1, y=y, p3=p3, s=s))
        x = x + v
    E, hs = head_forward(x, batch["idx_m"], int(batch["n_mol"]), hp)
    gx = head_backward_R(hs, hp)                                            # ---- pass B
    gd = torch.zeros_like(g["d"])
    for l in reversed(range(L)):
        pre, S = "interactions.%d." % l, lay[l]
        gs = gx @ p[pre + "f2out.1.weight"]
        gy = (gs * act_order("ssp", 1, S["p3"])) @ p[pre + "f2out.0.weight"]
        gd = gd + (gy[ii] * S["h"][jj] * S["Wf1"]).sum(1)
        if l > 0:
            gh = seg_sum(gy[ii] * S["Wf"], jj, N)
            gx = gx + gh @ p[pre + "in2f.weight"]
    F = forces_from_edge_gradient(g, gd)
    return E, F, dict(p=p, hp=hp, g=g, lay=lay, x_out=x, hs=hs, L=L, batch=batch)


def schnet_backward(saved, gE: Tensor, gF: Tensor) -> Dict[str, Tensor]:
    """Passes C + D: gradients of L w.r.t. every weight given gE = dL/dE [M], gF = dL/dF [N,3]."""
    p, hp, g, lay, L, batch = (saved[k] for k in ("p", "hp", "g", "lay", "L", "batch"))
    ii, jj, N = g["ii"], g["jj"], g["N"]
    dtype = g["d"].dtype
    dt, _ = tangent_geometry(g, -gF.to(dtype))
    xt = torch.zeros_like(lay[0]["x"])
    for l in range(L):                                                      # ---- pass C
        pre, S = "interactions.%d." % l, lay[l]
        ht = xt @ p[pre + "in2f.weight"].t()
        yt = seg_sum(ht[jj] * S["Wf"] + S["h"][jj] * S["Wf1"] * dt[:, None], ii, N)
        p3t = yt @ p[pre + "f2out.0.weight"].t()
        st = act_order("ssp", 1, S["p3"]) * p3t
        S.update(xt=xt, ht=ht, yt=yt, p3t=p3t, st=st)
        xt = xt + st @ p[pre + "f2out.1.weight"].t()
    grads: Dict[str, Tensor] = {}
    gx, hx = head_dual_backward(saved["x_out"], xt, saved["hs"], hp, gE.to(dtype)[batch["idx_m"]], grads)
    for l in reversed(range(L)):                                            # ---- pass D
        pre, S = "interactions.%d." % l, lay[l]
        w4, w3, w2, w1, win = (p[pre + k] for k in ("f2out.1.weight", "f2out.0.weight", "filter_network.1.weight",
                                                     "filter_network.0.weight", "in2f.weight"))
        grads[pre + "f2out.1.weight"] = gx.t() @ S["s"] + hx.t() @ S["st"]
        grads[pre + "f2out.1.bias"] = gx.sum(0)
        gs, hs_ = gx @ w4, hx @ w4
        gp = gs * act_order("ssp", 1, S["p3"]) + hs_ * act_order("ssp", 2, S["p3"]) * S["p3t"]
        hp_ = hs_ * act_order("ssp", 1, S["p3"])
        grads[pre + "f2out.0.weight"] = gp.t() @ S["y"] + hp_.t() @ S["yt"]
        grads[pre + "f2out.0.bias"] = gp.sum(0)
        gy, hy = gp @ w3, hp_ @ w3
        # y_i = sum h_j Wf_e ;  yt_i = sum (ht_j Wf_e + h_j Wf1_e dt_e)
        gh = seg_sum(gy[ii] * S["Wf"] + hy[ii] * S["Wf1"] * dt[:, None], jj, N)
        hh = seg_sum(hy[ii] * S["Wf"], jj, N)
        gWf = gy[ii] * S["h"][jj] + hy[ii] * S["ht"][jj]
        Q = hy[ii] * S["h"][jj] * dt[:, None]                                # cotangent of Wf1
        gg, gg1 = gWf * g["fc"][:, None] + Q * g["fc1"][:, None], Q * g["fc"][:, None]
        grads[pre + "filter_network.1.weight"] = gg.t() @ S["z"] + gg1.t() @ S["z1"]
        grads[pre + "filter_network.1.bias"] = gg.sum(0)
        gz, gz1 = gg @ w2, gg1 @ w2
        ga = gz * act_order("ssp", 1, S["a"]) + gz1 * act_order("ssp", 2, S["a"]) * S["a1"]
        ga1 = gz1 * act_order("ssp", 1, S["a"])
        grads[pre + "filter_network.0.weight"] = ga.t() @ g["phi"] + ga1.t() @ g["phi1"]
        grads[pre + "filter_network.0.bias"] = ga.sum(0)
        grads[pre + "in2f.weight"] = gh.t() @ S["x"] + hh.t() @ S["xt"]
        gx, hx = gx + gh @ win, hx + hh @ win
    grads["embedding.weight"] = seg_sum(gx, batch["Z"], p["embedding.weight"].shape[0])
    return grads


# ================================================================================================ PaiNN
def _split3(x, F):
    return x[..., :F], x[..., F:2 * F], x[..., 2 * F:]


def painn_forward(rep_p, head_p, batch, L: int, dtype=torch.float64, shared_filters: bool = False, eps: float = 1e-8):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in rep_p.items()}
    hp = {k: v.to(dtype) for k, v in head_p.items()}
    g = geometry(batch, p, dtype)
    ii, jj, N, u = g["ii"], g["jj"], g["N"], g["u"]
    q = p["embedding.weight"][batch["Z"]]
    F = q.shape[1]
    mu = torch.zeros((N, 3, F), dtype=dtype)
    Praw, Praw1 = g["phi"] @ p["filter_net.weight"].t() + p["filter_net.bias"], g["phi1"] @ p["filter_net.weight"].t()
    Phi_all = Praw * g["fc"][:, None]
    Phi1_all = Praw1 * g["fc"][:, None] + Praw * g["fc1"][:, None]
    lay = []
    for l in range(L):                                                      # ---- pass A
        pi, pm = "interactions.%d.interatomic_context_net." % l, "mixing.%d." % l
        sl = slice(0, 3 * F) if shared_filters else slice(3 * F * l, 3 * F * (l + 1))
        Phi, Phi1 = Phi_all[:, sl], Phi1_all[:, sl]
        pa = q @ p[pi + "0.weight"].t() + p[pi + "0.bias"]
        sa = act_order("silu", 0, pa)
        c = sa @ p[pi + "1.weight"].t() + p[pi + "1.bias"]
        m = Phi * c[jj]
        mq, mR, mm = _split3(m, F)
        q1 = q + seg_sum(mq, ii, N)
        mu1 = mu + seg_sum(mR[:, None, :] * u[:, :, None] + mm[:, None, :] * mu[jj], ii, N)
        VW = mu1 @ p[pm + "mu_channel_mix.weight"].t()
        V, W = VW[..., :F], VW[..., F:]
        n = torch.sqrt((V * V).sum(1) + eps)
        ctx = torch.cat([q1, n], 1)
        pb = ctx @ p[pm + "intraatomic_context_net.0.weight"].t() + p[pm + "intraatomic_context_net.0.bias"]
        sb = act_order("silu", 0, pb)
        a = sb @ p[pm + "intraatomic_context_net.1.weight"].t() + p[pm + "intraatomic_context_net.1.bias"]
        aq, am, aqm = _split3(a, F)
        svw = (V * W).sum(1)
        lay.append(dict(q=q, mu=mu, pa=pa, sa=sa, c=c, Phi=Phi, Phi1=Phi1, sl=sl, q1=q1, mu1=mu1, V=V, W=W, n=n, ctx=ctx, pb=pb, sb=sb,
                        a=a, svw=svw))
        q = q1 + aq + aqm * svw
        mu = mu1 + am[:, None, :] * W
    E, hs = head_forward(q, batch["idx_m"], int(batch["n_mol"]), hp)
    gq = head_backward_R(hs, hp)                                            # ---- pass B
    gmu = torch.zeros_like(mu)
    gd = torch.zeros_like(g["d"])
    gu = torch.zeros_like(u)
    for l in reversed(range(L)):
        pi, pm, S = "interactions.%d.interatomic_context_net." % l, "mixing.%d." % l, lay[l]
        aq, am, aqm = _split3(S["a"], F)
        V, W = S["V"], S["W"]
        # mixing
        ga = torch.cat([gq, (gmu * W).sum(1), gq * S["svw"]], 1)
        gsv = gq * aqm
        gW = gmu * am[:, None, :] + gsv[:, None, :] * V
        gV = gsv[:, None, :] * W
        gpb = (ga @ p[pm + "intraatomic_context_net.1.weight"]) * act_order("silu", 1, S["pb"])
        gctx = gpb @ p[pm + "intraatomic_context_net.0.weight"]
        gq1 = gq + gctx[:, :F]
        gV = gV + (gctx[:, F:] / S["n"])[:, None, :] * V
        gmu1 = gmu + torch.cat([gV, gW], 2) @ p[pm + "mu_channel_mix.weight"]
        # message
        cj, muj = S["c"][jj], S["mu"][jj]
        gm = torch.cat([gq1[ii], (gmu1[ii] * u[:, :, None]).sum(1), (gmu1[ii] * muj).sum(1)], 1)
        gd = gd + (gm * cj * S["Phi1"]).sum(1)
        mR, mm = (S["Phi"] * cj)[:, F:2 * F], (S["Phi"] * cj)[:, 2 * F:]
        gu = gu + (gmu1[ii] * mR[:, None, :]).sum(2)
        gq, gmu = gq1, gmu1
        if l > 0:
            gc = seg_sum(S["Phi"] * gm, jj, N)
            gmu = gmu + seg_sum(mm[:, None, :] * gmu1[ii], jj, N)
            gpa = (gc @ p[pi + "1.weight"]) * act_order("silu", 1, S["pa"])
            gq = gq + gpa @ p[pi + "0.weight"]
    Fo = forces_from_edge_gradient(g, gd, gu)
    return E, Fo, dict(p=p, hp=hp, g=g, lay=lay, x_out=q, hs=hs, L=L, batch=batch, F=F, eps=eps, shared=shared_filters)


def painn_backward(saved, gE: Tensor, gF: Tensor) -> Dict[str, Tensor]:
    p, hp, g, lay, L, batch, F = (saved[k] for k in ("p", "hp", "g", "lay", "L", "batch", "F"))
    ii, jj, N, u = g["ii"], g["jj"], g["N"], g["u"]
    dtype = g["d"].dtype
    dt, ut = tangent_geometry(g, -gF.to(dtype))
    qt = torch.zeros_like(lay[0]["q"])
    mut = torch.zeros_like(lay[0]["mu"])
    for l in range(L):                                                      # ---- pass C
        pi, pm, S = "interactions.%d.interatomic_context_net." % l, "mixing.%d." % l, lay[l]
        pat = qt @ p[pi + "0.weight"].t()
        sat = act_order("silu", 1, S["pa"]) * pat
        ct = sat @ p[pi + "1.weight"].t()
        cj, ctj = S["c"][jj], ct[jj]
        m = S["Phi"] * cj
        mt = S["Phi1"] * dt[:, None] * cj + S["Phi"] * ctj
        _, mR, mm = _split3(m, F)
        mqt, mRt, mmt = _split3(mt, F)
        q1t = qt + seg_sum(mqt, ii, N)
        mu1t = mut + seg_sum(mRt[:, None, :] * u[:, :, None] + mR[:, None, :] * ut[:, :, None] + mmt[:, None, :] * S["mu"][jj]
                             + mm[:, None, :] * mut[jj], ii, N)
        VWt = mu1t @ p[pm + "mu_channel_mix.weight"].t()
        Vt, Wt = VWt[..., :F], VWt[..., F:]
        nt = (S["V"] * Vt).sum(1) / S["n"]
        ctxt = torch.cat([q1t, nt], 1)
        pbt = ctxt @ p[pm + "intraatomic_context_net.0.weight"].t()
        sbt = act_order("silu", 1, S["pb"]) * pbt
        at = sbt @ p[pm + "intraatomic_context_net.1.weight"].t()
        aq, am, aqm = _split3(S["a"], F)
        aqt, amt, aqmt = _split3(at, F)
        svwt = (Vt * S["W"] + S["V"] * Wt).sum(1)
        S.update(qt=qt, mut=mut, pat=pat, sat=sat, ct=ct, q1t=q1t, mu1t=mu1t, Vt=Vt, Wt=Wt, nt=nt, ctxt=ctxt, pbt=pbt, sbt=sbt, at=at,
                 svwt=svwt)
        qt = q1t + aqt + aqmt * S["svw"] + aqm * svwt
        mut = mu1t + amt[:, None, :] * S["W"] + am[:, None, :] * Wt
    grads: Dict[str, Tensor] = {}
    gq, hq = head_dual_backward(saved["x_out"], qt, saved["hs"], hp, gE.to(dtype)[batch["idx_m"]], grads)
    gmu, hmu = torch.zeros_like(mut), torch.zeros_like(mut)
    gWf = torch.zeros_like(p["filter_net.weight"])
    gbf = torch.zeros_like(p["filter_net.bias"])
    for l in reversed(range(L)):                                            # ---- pass D
        pi, pm, S = "interactions.%d.interatomic_context_net." % l, "mixing.%d." % l, lay[l]
        aq, am, aqm = _split3(S["a"], F)
        aqt, amt, aqmt = _split3(S["at"], F)
        V, W, Vt, Wt, n = S["V"], S["W"], S["Vt"], S["Wt"], S["n"]
        # ---- mixing (painn.py:99-116), equations (1)-(8) of DESIGN 4.12
        ga = torch.cat([gq, (gmu * W + hmu * Wt).sum(1), gq * S["svw"] + hq * S["svwt"]], 1)
        ha = torch.cat([hq, (hmu * W).sum(1), hq * S["svw"]], 1)
        gs, hs_ = gq * aqm + hq * aqmt, hq * aqm
        gW = gmu * am[:, None, :] + hmu * amt[:, None, :] + gs[:, None, :] * V + hs_[:, None, :] * Vt
        hW = hmu * am[:, None, :] + hs_[:, None, :] * V
        gV = gs[:, None, :] * W + hs_[:, None, :] * Wt
        hV = hs_[:, None, :] * W
        wb2, wb1 = p[pm + "intraatomic_context_net.1.weight"], p[pm + "intraatomic_context_net.0.weight"]
        grads[pm + "intraatomic_context_net.1.weight"] = ga.t() @ S["sb"] + ha.t() @ S["sbt"]
        grads[pm + "intraatomic_context_net.1.bias"] = ga.sum(0)
        gsb, hsb = ga @ wb2, ha @ wb2
        gpb = gsb * act_order("silu", 1, S["pb"]) + hsb * act_order("silu", 2, S["pb"]) * S["pbt"]
        hpb = hsb * act_order("silu", 1, S["pb"])
        grads[pm + "intraatomic_context_net.0.weight"] = gpb.t() @ S["ctx"] + hpb.t() @ S["ctxt"]
        grads[pm + "intraatomic_context_net.0.bias"] = gpb.sum(0)
        gctx, hctx = gpb @ wb1, hpb @ wb1
        gq1, hq1 = gq + gctx[:, :F], hq + hctx[:, :F]
        gn, hn = gctx[:, F:], hctx[:, F:]
        gV = gV + (gn / n)[:, None, :] * V + (hn / n)[:, None, :] * (Vt - (S["nt"] / n)[:, None, :] * V)
        hV = hV + (hn / n)[:, None, :] * V
        wmix = p[pm + "mu_channel_mix.weight"]
        gVW, hVW = torch.cat([gV, gW], 2), torch.cat([hV, hW], 2)
        grads[pm + "mu_channel_mix.weight"] = (gVW.reshape(-1, 2 * F).t() @ S["mu1"].reshape(-1, F)
                                               + hVW.reshape(-1, 2 * F).t() @ S["mu1t"].reshape(-1, F))
        gmu1, hmu1 = gmu + gVW @ wmix, hmu + hVW @ wmix
        # ---- message (painn.py:50-66), equations (a)-(f)
        cj, ctj, muj, mutj = S["c"][jj], S["ct"][jj], S["mu"][jj], S["mut"][jj]
        gm = torch.cat([gq1[ii], (gmu1[ii] * u[:, :, None] + hmu1[ii] * ut[:, :, None]).sum(1), (gmu1[ii] * muj + hmu1[ii] * mutj).sum(1)], 1)
        hm = torch.cat([hq1[ii], (hmu1[ii] * u[:, :, None]).sum(1), (hmu1[ii] * muj).sum(1)], 1)
        m = S["Phi"] * cj
        mt = S["Phi1"] * dt[:, None] * cj + S["Phi"] * ctj
        mm, mmt = m[:, 2 * F:], mt[:, 2 * F:]
        gPhi = gm * cj + hm * ctj
        gPhi1 = hm * cj * dt[:, None]
        gPraw = gPhi * g["fc"][:, None] + gPhi1 * g["fc1"][:, None]
        gPraw1 = gPhi1 * g["fc"][:, None]
        gWf[S["sl"]] += gPraw.t() @ g["phi"] + gPraw1.t() @ g["phi1"]
        gbf[S["sl"]] += gPraw.sum(0)
        gc = seg_sum(gm * S["Phi"] + hm * S["Phi1"] * dt[:, None], jj, N)
        hc = seg_sum(hm * S["Phi"], jj, N)
        gq, hq = gq1, hq1
        gmu = gmu1 + seg_sum(gmu1[ii] * mm[:, None, :] + hmu1[ii] * mmt[:, None, :], jj, N)
        hmu = hmu1 + seg_sum(hmu1[ii] * mm[:, None, :], jj, N)
        wa2, wa1 = p[pi + "1.weight"], p[pi + "0.weight"]
        grads[pi + "1.weight"] = gc.t() @ S["sa"] + hc.t() @ S["sat"]
        grads[pi + "1.bias"] = gc.sum(0)
        gsa, hsa = gc @ wa2, hc @ wa2
        gpa = gsa * act_order("silu", 1, S["pa"]) + hsa * act_order("silu", 2, S["pa"]) * S["pat"]
        hpa = hsa * act_order("silu", 1, S["pa"])
        grads[pi + "0.weight"] = gpa.t() @ S["q"] + hpa.t() @ S["qt"]
        grads[pi + "0.bias"] = gpa.sum(0)
        gq, hq = gq + gpa @ wa1, hq + hpa @ wa1
    grads["filter_net.weight"], grads["filter_net.bias"] = gWf, gbf
    grads["embedding.weight"] = seg_sum(gq, batch["Z"], p["embedding.weight"].shape[0])
    return grads
