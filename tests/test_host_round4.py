"""Host logic of round 4 (no GPU): the install() defaults keep the reference classes scriptable (round-3 ADVICE), the block plan's
size table, the C ABI exports of the block kernels."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference():
    from oracle import refshim
    if not refshim.available():
        pytest.skip("neither /root/reference nor oracle/_ref is present")
    ns = refshim.load()
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    return ns


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_reference_model_scripts_after_default_install(kind):
    """install() (fused head + the standard potential routed at ``__call__``) must leave ``NeuralNetworkPotential.forward`` the
    reference's own, scriptable function: spkdeploy (src/scripts/spkdeploy:16-40) and the LAMMPS pair style
    (interfaces/lammps/pair_schnetpack.cpp:125-131) script / load exactly this class."""
    ns = _reference()
    import schnetpack_amd.install as inst
    spk = sys.modules["schnetpack"]
    fwd_before = ns.model.NeuralNetworkPotential.forward
    inst.install(spk)
    try:
        cls = ns.model.NeuralNetworkPotential
        assert cls.forward is fwd_before and getattr(cls.__call__, "_spk_hip_patched", False)
        rb, cf = spk.nn.GaussianRBF(20, 5.0), spk.nn.CosineCutoff(5.0)
        rep_cls = getattr(sys.modules["schnetpack.representation." + kind], "SchNet" if kind == "schnet" else "PaiNN")
        aw = sys.modules["schnetpack.atomistic.atomwise"].Atomwise(n_in=128, output_key="energy")
        pd = sys.modules["schnetpack.atomistic.distances"].PairwiseDistances()
        m = cls(rep_cls(128, 3, rb, cf), input_modules=[pd], output_modules=[aw, ns.response.Forces()]).eval()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scripted = torch.jit.script(m)
        assert isinstance(scripted, torch.jit.ScriptModule)
        assert "spk_hip" in str(scripted.representation.graph) or "spk_hip" in str(scripted.inlined_graph)     # the HIP operators are in the archive
        # a model with forward hooks, or called with keyword arguments, takes nn.Module.__call__ (hooks are honoured)
        seen = []
        h = m.register_forward_pre_hook(lambda mod, args: seen.append(1))
        with pytest.raises(Exception):
            m({})          # the reference forward runs (and fails on the empty batch) -- after the hook fired
        h.remove()
        assert seen == [1]
    finally:
        inst.uninstall()
    assert "__call__" not in ns.model.NeuralNetworkPotential.__dict__


def test_block_plan_size_table_and_exports():
    from schnetpack_amd import _lib
    L = _lib.lib()
    BA = int(L.spk_blocks_group_atoms())
    assert BA in (8, 16)
    sizes = (ctypes.c_int64 * 12)()
    _lib.check(L.spk_blocks_sizes(31944, 1711014, 20, 128, sizes))
    ng = (31944 + BA - 1) // BA
    assert sizes[0] == ng and sizes[1] == ng * BA and sizes[2] == 1711014 and sizes[3] == 1711014 and sizes[4] == 31945
    assert sizes[9] == 5 and sizes[8] == 8 * 4 * 1711014 and sizes[10] >= 4 * ng * BA
    _lib.check(L.spk_blocks_sizes(10, 0, 25, 64, sizes))
    assert sizes[9] == 8 and sizes[2] == 1
    assert L.spk_blocks_sizes(-1, 0, 20, 128, sizes) != 0
    for name in ("spk_blocks_build", "spk_blocks_prepare_f32", "spk_painn_set_block", "spk_painn_blk_set_debug_buffer"):
        assert hasattr(L, name)


def test_flat_adamw_chunk_table_and_cpu_refusal():
    """train.FlatAdamW: the chunk table covers every parameter element exactly once in bucket order (chunks of at most 2048), and the
    optimizer has no CPU route -- stepping on host tensors fails loudly (the product path never falls back)."""
    import pytest
    import torch
    from schnetpack_amd._lib import SpkHipError
    from schnetpack_amd.parallel import FlatGradAllReduce
    from schnetpack_amd.train import FlatAdamW
    shapes = [(128, 20), (128,), (3, 50, 50), (1,), (2049,)]
    params = [torch.nn.Parameter(torch.randn(*sh)) for sh in shapes]
    red = FlatGradAllReduce(params, as_views=True)
    opt = FlatAdamW(red, lr=1e-3)
    rows = opt.chunks.tolist()
    assert opt.numel == sum(p.numel() for p in params) == red.flat.numel()
    flat_off = 0
    for p in params:
        mine = [r for r in rows if r[0] == p.data_ptr()]
        assert [r[1] for r in mine] == list(range(0, p.numel(), FlatAdamW.CHUNK))          # offsets inside the parameter
        assert [r[2] for r in mine] == [flat_off + c for c in range(0, p.numel(), FlatAdamW.CHUNK)]   # offsets inside the bucket
        assert sum(r[3] for r in mine) == p.numel() and all(0 < r[3] <= FlatAdamW.CHUNK for r in mine)
        flat_off += p.numel()
    assert len(rows) == sum((p.numel() + FlatAdamW.CHUNK - 1) // FlatAdamW.CHUNK for p in params)
    with pytest.raises(SpkHipError):
        opt.step()
    with pytest.raises(ValueError):
        FlatAdamW(FlatGradAllReduce([torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))], as_views=False))


def test_filter_table_registry_stamps():
    """Host side of the tabulated-filter registry (spk_filter_table_set / _set_stamp / _drop_if_stale; pointers are only stored, nothing is
    launched): a table whose recorded weight version differs from the caller's is dropped, an untracked or matching one stays."""
    import ctypes
    from schnetpack_amd import _lib
    L = _lib.lib()
    key, tab = ctypes.c_void_p(0x7000_0000_1000), ctypes.c_void_p(0x7000_0000_2000)
    L.spk_filter_table_clear()
    try:
        assert L.spk_filter_table_set_stamp(key, 5) != 0                      # nothing registered under this key
        _lib.check(L.spk_filter_table_set(key, tab, 512, 5.0))
        assert L.spk_filter_table_drop_if_stale(key, 123) == 0                # untracked (stamp 0): never dropped
        _lib.check(L.spk_filter_table_set_stamp(key, 7))
        assert L.spk_filter_table_drop_if_stale(key, 7) == 0                  # same version: stays
        assert L.spk_filter_table_drop_if_stale(ctypes.c_void_p(0x7000_0000_3000), 9) == 0      # other key: untouched
        assert L.spk_filter_table_drop_if_stale(key, 8) == 1                  # the weights moved on: dropped
        assert L.spk_filter_table_drop_if_stale(key, 8) == 0                  # ... and gone
        _lib.check(L.spk_filter_table_set(key, tab, 512, 5.0))                # re-registering resets the stamp
        assert L.spk_filter_table_drop_if_stale(key, 99) == 0
    finally:
        L.spk_filter_table_clear()
