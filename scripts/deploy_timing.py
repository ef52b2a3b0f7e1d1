"""PCIe-inclusive rate of the torch-free runtime (host arrays in, host energy / forces out) on the two bench
workloads, next to the resident-in-HBM graph replay bench.py reports.  Prints one JSON object."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from schnetpack_amd import deploy, model as M, synthetic as S  # noqa: E402
from schnetpack_amd.csrc import build as B  # noqa: E402

out = {}
tmp = tempfile.mkdtemp()
for kind in ("schnet", "painn"):
    torch.manual_seed(0)
    m = M.build_model(kind, 128, 3).eval()
    path = os.path.join(tmp, kind + ".spkm")
    deploy.export_potential(m, path)
    pot = deploy.DeployedPotential(path)
    for wl in ("aspirin256", "water32k"):
        if wl == "aspirin256":
            b = S.molecule_batch("aspirin", 256, seed=0)
            cell = pbc = None
        else:
            b = S.water_box()
            cell, pbc = b["cell"].numpy().reshape(1, 3, 3), np.ones((1, 3), np.uint8)
        n_mol = int(b["n_mol"])
        Z, R, im = b["Z"].numpy(), b["R"].numpy().astype(np.float32), b["idx_m"].numpy()
        ii, jj, off = b["idx_i"].numpy(), b["idx_j"].numpy(), b["offsets"].numpy().astype(np.float32)
        E = int(ii.shape[0])
        rec = {"n_atoms": int(Z.shape[0]), "n_edges": E}
        for mode in ("explicit_list", "device_list_skin0", "device_list_skin2_reused"):
            def call():
                if mode == "explicit_list":
                    return pot.compute(Z, R, ii, jj, off, im, n_mol)
                skin = 0.0 if mode.endswith("skin0") else 2.0
                return pot.compute_cell(Z, R, cell, pbc, im, n_mol, skin=skin)
            for _ in range(3):
                call()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                call()
            ms = (time.perf_counter() - t0) / reps * 1e3
            rec[mode] = {"ms_per_call": round(ms, 4), "M_edge_messages_per_s": round(E * 3 / ms / 1e3, 1)}
        out[kind + "_" + wl] = rec
        print(kind, wl, rec, file=sys.stderr)
print(json.dumps(out))
