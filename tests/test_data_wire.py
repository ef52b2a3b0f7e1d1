"""Collate-side wire format (schnetpack_amd/data.py): the collate mirror equals the reference's
``_atoms_collate_fn`` (data/loader.py:13-58) and the host-made plan equals what ``spk_edge_plan`` derives on the device."""
import numpy as np
import pytest
import torch

from oracle import refshim
from schnetpack_amd import data as D, synthetic as S


def _systems(seed=0, kinds=("aspirin", "ethanol", "aspirin", "ethanol", "ethanol")):
    rng = np.random.RandomState(seed)
    out = []
    for kind in kinds:
        Z, R0 = (S.ASPIRIN_Z, np.asarray(S.ASPIRIN_R)) if kind == "aspirin" else (S.ETHANOL_Z, np.asarray(S.ETHANOL_R))
        R = R0 + 0.05 * rng.randn(*R0.shape)
        ii, jj = S.neighbor_pairs_open(R, 5.0)
        n = len(Z)
        out.append({"_atomic_numbers": torch.tensor(Z), "_positions": torch.from_numpy(R).float(), "_n_atoms": torch.tensor([n]),
                    "_idx_i": torch.from_numpy(ii), "_idx_j": torch.from_numpy(jj), "_offsets": torch.zeros(len(ii), 3),
                    "_cell": torch.zeros(1, 3, 3), "_pbc": torch.zeros(3, dtype=torch.bool), "energy": torch.randn(1),
                    "_idx_i_triples": torch.arange(min(4, len(ii))), "_idx_j_triples": torch.arange(min(4, len(ii))),
                    "_idx_k_triples": torch.arange(min(4, len(ii)))[::1].clone()})
    return out


@pytest.mark.skipif(not refshim.available(), reason="reference not present")
def test_collate_mirror_equals_reference_collate():
    ns = refshim.load()
    batch = _systems()
    ref = ns.loader._atoms_collate_fn(batch)
    got = D.atoms_collate_fn(batch)
    assert set(ref) == set(got)
    for k in ref:
        assert ref[k].dtype == got[k].dtype and torch.equal(ref[k], got[k]), k


def _brute_plan(ii, jj, off, n):
    E = len(ii)
    rev = np.full(E, -1)
    for e in range(E):
        for f in range(E):
            if ii[f] == jj[e] and jj[f] == ii[e] and np.array_equal(off[f], -off[e]):
                rev[e] = f
    return rev


def test_host_plan_against_brute_force_and_degenerate_lists():
    b = D.WireCollate(model_cutoff=3.5)(_systems(1, ("ethanol", "aspirin", "ethanol")))
    ii, jj, off = b["_idx_i"].numpy(), b["_idx_j"].numpy(), b["_offsets"].numpy()
    n = int(b["_atomic_numbers"].shape[0])
    meta = b["_spk_plan_meta"].tolist()
    assert meta[0] == 1 and meta[1] == 1 and 2 * meta[2] == len(ii)
    rev = b["_spk_rev"].numpy()
    assert np.array_equal(rev, _brute_plan(ii, jj, off, n))
    half = b["_spk_half"].numpy()
    assert np.all(rev[half] > half) and np.all(np.diff(half) > 0)
    ep = b["_spk_edge_pair"].numpy()
    assert np.array_equal(ep[half], np.arange(len(half))) and np.array_equal(ep[rev[half]], np.arange(len(half)))
    rp = b["_spk_rowptr"].numpy()
    assert rp[0] == 0 and rp[-1] == len(ii) and np.array_equal(np.diff(rp), np.bincount(ii, minlength=n))
    # groups: ethanol (9) + aspirin (21) = 30 <= 32 merge; the third molecule opens a new group
    assert b["_spk_grp_atom0"].tolist() == [0, 30, 39] and meta[3] == 2 and meta[4] == 30
    assert meta[6] == 1                         # list built with 5 A, model cutoff 3.5 A: > 5 % of the pairs are outside
    # periodic images: the same (i, j) twice with opposite shifts pairs up by the shift
    ii2 = np.array([0, 0, 1, 1]); jj2 = np.array([1, 1, 0, 0])
    off2 = np.array([[0, 0, 0], [3.0, 0, 0], [0, 0, 0], [-3.0, 0, 0]], dtype=np.float32)
    p2 = D.host_plan(ii2, jj2, off2, 2)
    assert p2["meta"][1] == 1 and p2["rev"].tolist() == [2, 3, 0, 1]
    # asymmetric and unsorted lists are flagged, not paired
    assert D.host_plan(np.array([0, 1, 1]), np.array([1, 0, 2]), None, 3)["meta"][1] == 0
    assert D.host_plan(np.array([1, 0]), np.array([0, 1]), None, 2)["meta"][0] == 0
    with pytest.raises(ValueError):
        D.host_plan(np.array([0]), np.array([5]), None, 2)


def test_loader_runs_the_wire_collate_in_workers():
    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.items = _systems(2, ("aspirin",) * 6)

        def __len__(self):
            return len(self.items)

        def __getitem__(self, k):
            return self.items[k]
    dl = D.AtomsLoader(DS(), batch_size=3, num_workers=2)
    batches = list(dl)
    assert len(batches) == 2 and all("_spk_rowptr" in b and b["_spk_plan_meta"][1] == 1 for b in batches)
    assert batches[0]["_idx_m"].tolist() == [0] * 21 + [1] * 21 + [2] * 21


@pytest.mark.gpu
def test_installed_host_plan_equals_device_plan_and_force_call():
    """On the device: the installed plan is bit-identical to what spk_edge_plan derives, and the force call through it matches."""
    from schnetpack_amd import model as M
    from conftest import rel_err
    dev = torch.device("cuda:0")
    batch = D.WireCollate(model_cutoff=5.0)(_systems(3, ("aspirin", "ethanol", "aspirin", "aspirin", "ethanol", "ethanol")))
    torch.manual_seed(0)
    model = M.build_model("schnet").to(dev).eval()
    inp = D.to_device({k: v for k, v in batch.items() if k not in ("energy",) and "triples" not in k and not k.endswith("_local")}, dev)
    assert inp["_spk_plan_meta"].device.type == "cpu"          # the host reads it: no device-to-host copy per batch
    with pytest.raises(ValueError, match="keep it on the host"):
        D.install_plan({k: v.to(dev) for k, v in inp.items()})
    bad = dict(inp)
    bad["_spk_rev"] = inp["_spk_rev"] + 10 ** 6
    with pytest.raises(ValueError, match="rev outside"):
        D.install_plan(bad, validate=True)
    assert D.install_plan(inp, validate=True)
    N = int(inp["_atomic_numbers"].shape[0])
    out = model({k: v for k, v in inp.items() if not k.startswith("_spk")})
    r = inp["_positions"][inp["_idx_j"]] - inp["_positions"][inp["_idx_i"]] + inp["_offsets"]
    got = torch.ops.spk_hip.edge_plan(inp["_idx_i"], inp["_idx_j"], N, r)          # cache hit: the installed plan
    fresh_i, fresh_j = inp["_idx_i"].clone(), inp["_idx_j"].clone()                 # new tensors: derived on the device
    ref = torch.ops.spk_hip.edge_plan(fresh_i, fresh_j, N, r, 5.0)
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a, b)
    assert got[3][:3].tolist() == ref[3][:3].tolist() and int(got[3][3]) == 0 == int(ref[3][3])
    inp2 = {k: v for k, v in inp.items() if not k.startswith("_spk")}
    inp2["_idx_i"], inp2["_idx_j"] = fresh_i, fresh_j
    out2 = model(inp2)
    assert rel_err(out["forces"].detach().cpu(), out2["forces"].detach().cpu()) < 2e-6
    assert rel_err(out["energy"].detach().cpu(), out2["energy"].detach().cpu()) < 2e-6
