import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    """Raw arrays of a fixture file."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (z[k].item() if z[k].shape == () else z[k]) for k in z.files}


def load_golden(name):
    """Return (batch dict of tensors, reference outputs dict of tensors, meta dict)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    batch, ref, meta, weights = {}, {}, {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith("in_"):
            batch[k[3:]] = int(v) if v.shape == () else torch.from_numpy(v)
        elif k.startswith("ref_"):
            ref[k[4:]] = torch.from_numpy(v)
        elif k.startswith("w_"):
            weights[k[2:]] = torch.from_numpy(v)
        else:
            meta[k] = v.item() if v.shape == () else v
    meta["weights"] = weights
    return batch, ref, meta


def golden_params(meta):
    """Representation + head parameters of a golden case (seeded init or stored)."""
    from oracle import spk_oracle as O
    if meta["weights"]:
        rep = {k[4:]: v for k, v in meta["weights"].items() if k.startswith("rep.")}
        head = {k[5:]: v for k, v in meta["weights"].items() if k.startswith("head.")}
        return rep, head
    kw = dict(cutoff=float(meta["cutoff"]), radial=str(meta["radial"]), n_interactions=int(meta.get("n_interactions", 3)))
    if str(meta["kind"]) == "schnet":
        rep = O.init_schnet_params(**kw)
    else:
        rep = O.init_painn_params(**kw)
    return rep, O.init_atomwise_params(128, seed=1)


MODEL_CASES = ["schnet_ethanol.npz", "schnet_aspirin8.npz", "painn_ethanol.npz",
               "painn_aspirin8.npz", "schnet_bessel_aspirin2.npz", "painn_bessel_aspirin2.npz",
               "schnet_skin_aspirin2.npz", "painn_skin_aspirin2.npz",
               "painn_aspirin_pretrained.npz", "schnet_water192.npz", "painn_water192.npz",
               # the reference's default depth (6 interactions): twice the error accumulation of the bench configuration
               "schnet6_aspirin4.npz", "schnet6_water192.npz", "painn6_aspirin4.npz", "painn6_water192.npz",
               # six interactions x BesselRBF: hardware sin / cos in the molecule kernels x depth, the least-margin combination (round-3 review)
               "schnet6_bessel_aspirin4.npz", "schnet6_bessel_water192.npz", "painn6_bessel_aspirin4.npz", "painn6_bessel_water192.npz"]


def rel_err(a, b):
    """max|a-b| / max|b| -- the north_star's 'relative' for energies / forces."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
