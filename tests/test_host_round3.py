"""Host logic added in round 3 (no GPU): routing of the standard potential, the wire format's host-resident plan meta, the
bookkeeping of the bench line."""
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_classify_potential_routes_both_models_and_leaves_other_compositions():
    from schnetpack_amd import model as M
    from schnetpack_amd.atomistic import Atomwise, Forces, PairwiseDistances
    for kind in ("schnet", "painn"):
        m = M.build_model(kind)
        assert M.classify_potential(m) == 2 and m._potential_forces          # energies and forces: the two-launch operator
        m.output_modules[1].calc_stress = True
        assert M.classify_potential(m) == 0                                  # stress wanted: the reference composition runs
    # energy head only (no Forces module): SchNet has the differentiable one-operator form, PaiNN keeps module by module
    rep = M.build_model("schnet").representation
    only_e = M.NeuralNetworkPotential(rep, input_modules=[PairwiseDistances()], output_modules=[Atomwise(n_in=128, output_key="energy")])
    assert M.classify_potential(only_e) == 1
    rep_p = M.build_model("painn").representation
    only_e = M.NeuralNetworkPotential(rep_p, input_modules=[PairwiseDistances()], output_modules=[Atomwise(n_in=128, output_key="energy")])
    assert M.classify_potential(only_e) == 0
    # an averaged energy cannot take the forces form (its Forces module differentiates the mean)
    m = M.build_model("schnet")
    m.output_modules[0].aggregation_mode = "avg"
    assert M.classify_potential(m) == 1


def test_to_device_keeps_the_plan_meta_on_the_host_and_install_refuses_a_moved_one():
    from schnetpack_amd import data as D
    batch = {"_idx_i": torch.zeros(4, dtype=torch.int64), "_spk_plan_meta": torch.arange(8), "_spk_rowptr": torch.zeros(3, dtype=torch.int32)}
    moved = D.to_device(batch, torch.device("meta"))
    assert moved["_spk_plan_meta"].device.type == "cpu" and moved["_idx_i"].device.type == "meta"
    bad = dict(moved)
    bad["_spk_plan_meta"] = batch["_spk_plan_meta"].to("meta")
    with pytest.raises(ValueError, match="keep it on the host"):
        D.install_plan(bad)
    assert D.install_plan({"_idx_i": batch["_idx_i"]}) is False          # a batch without a host-made plan


def test_bench_books_every_modelled_kernel_with_a_lower_bound():
    B = _bench()
    for kind in ("schnet", "painn"):
        algo = B.algorithmic_work(kind, 77944, 5376, 256, 128, 3, 20)
        assert algo, kind
        for tag, (bound, work, executed, bmin) in algo.items():
            assert bound in ("mfma", "hbm") and work > 0 and 0 < executed <= 1.0, tag
            assert bmin is None or 0 < bmin, tag
    assert any(t.startswith("painn_mol") for t in B.algorithmic_work("painn", 77944, 5376, 256, 128, 3, 20))
    assert 0.05 <= B.RAMP_S <= 0.5          # the untimed clock ramp stays a bounded fraction of a second
