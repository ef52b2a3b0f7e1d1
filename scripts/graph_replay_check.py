import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import spk_oracle as O
from schnetpack_amd import _lib, model as M, synthetic as S
from schnetpack_amd.forcecall import GraphedForceCall
dev = torch.device("cuda:0")
head_p = O.init_atomwise_params(128, seed=1)
for kind in ("painn", "schnet"):
    rep_p = O.init_painn_params() if kind == "painn" else O.init_schnet_params()
    model = M.build_model(kind); M.load_reference_params(model, rep_p, head_p); model = model.to(dev).eval()
    for nmol in (6, 12, 20, 30, 40, 256):
        b = S.molecule_batch("aspirin", nmol, seed=3)
        inp = M.batch_to_inputs(b, dev)
        want = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()})
        wf = want["forces"].detach().clone()
        fc = GraphedForceCall(model)
        errs = []
        for it in range(4):
            got = fc(dict(inp)) if it < 1 else fc.replay()
            torch.cuda.synchronize()
            errs.append(float((got["forces"] - wf).abs().max() / wf.abs().max()))
        print(kind, "nmol", nmol, "N", b["Z"].shape[0], "errs", ["%.1e" % e for e in errs])
