"""``torch.ops.spk_hip.*`` -- the PyTorch-ROCm extension face of the HIP kernels.

``libspk_torch.so`` (``csrc/spk_torch.cpp``, built in-tree next to ``libspk_hip.so``) registers the operators with
``TORCH_LIBRARY``: C++ ``torch::autograd::Function`` wrappers, ROCm kernels (= calls into the C ABI of
``include/spk_hip.h`` on torch's current stream), Meta kernels (shape inference) and a loud CPU refusal.  The module
mirrors (``schnetpack_amd.nn``, ``.representation``, ``.atomistic``) call nothing else, which is what makes them
TorchScript-able: ``torch.jit.script(SchNet(...))`` (reference tests/nn/test_schnet.py:83-96), ``spkdeploy``
(src/scripts/spkdeploy:16-40), ``SchNetPackCalculator(script_model=True)``
(md/calculators/schnetpack_calculator.py:105-107).  From C++ (interfaces/lammps/pair_schnetpack.cpp:128) a scripted model
loads after ``dlopen("libspk_torch.so")``.

Importing this module loads both shared libraries; it raises ``SpkHipError`` when either has not been built -- there is
no eager / CPU fallback behind it.
"""
import os

import torch

from . import _lib
from ._lib import SpkHipError

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libspk_torch.so")
_loaded = False


def load():
    """Load libspk_hip.so, then the operator library (idempotent)."""
    global _loaded
    if _loaded:
        return torch.ops.spk_hip
    _lib.lib()      # RTLD_GLOBAL: libspk_torch.so resolves the C ABI from it
    if not os.path.exists(TORCH_LIB_PATH):
        raise SpkHipError("libspk_torch.so not found at %s -- build it with `python -m schnetpack_amd.csrc.build` "
                          "(there is no fallback path behind the torch.ops.spk_hip operators)" % TORCH_LIB_PATH)
    torch.ops.load_library(TORCH_LIB_PATH)
    _loaded = True
    return torch.ops.spk_hip


ops = load()

OPERATORS = ["scatter_add", "gather", "pairwise", "pairwise_backward", "dense", "radial_cutoff", "schnet", "painn", "atomwise",
             "dense_forward", "dense_backward_input", "radial_cutoff_backward", "schnet_forward", "schnet_backward", "painn_forward",
             "painn_backward", "atomwise_forward", "atomwise_backward", "schnet_potential", "schnet_potential_forward", "schnet_potential_backward", "schnet_potential_forces", "painn_potential_forces", "eval_guard", "potential_plan", "edge_plan", "edge_plan_install", "static_new", "static_release", "weights_changed", "static_declare", "static_declare_range", "static_refresh", "static_enable",
             "static_check", "static_clear", "clear_caches",
             # training regime: operators closed under differentiation (csrc/spk_torch_train.h)
             "act_mul", "linear", "matmul_nn", "matmul_tn", "cfconv", "edge_mul", "radial_d", "radial_c", "rowscale", "rowdot", "edge_norm", "vec3", "gemm_pair",
             "fm_loss", "fm_loss_forward", "fm_loss_backward"]


# operation codes of torch.ops.spk_hip.vec3 (include/spk_hip.h: SPK_VEC3_*)
VEC3_SCALE, VEC3_DOT, VEC3_OUTER, VEC3_CONTRACT, VEC3_ROWDOT = 0, 1, 2, 3, 4


class StaticLists:
    """Static-shape mode for HIP-graph replays of the differentiable (training) path.

    Plans of index tensors are normally cached on tensor identity / version and validated with a host round trip --
    neither survives a graph whose index BUFFERS are refilled between replays.  Inside ``with StaticLists() as sl:``
    (and in the captured graph) every index tensor declared with ``sl.declare_sorted(idx, n_rows)`` gets its CSR row
    pointers from a device-only kernel launched by ``sl.refresh()`` (capture that call at the start of the step); all
    other indices take the atomic scatter; neighbour-list plans (symmetry, reverse map) are not used.  ``sl.check()``
    polls the device flag that the refresh kernels raise when a declared index was not ascending / in range
    (``declare_range`` adds unsorted indices -- ``idx_j``, atomic numbers -- to that check)."""

    def __init__(self):
        # declarations are owned by THIS object: another StaticLists (a second shape bucket, a validation stepper) never frees
        # or re-purposes the row-pointer / error buffers a captured graph of this one points at (round-2 ADVICE)
        self._owner = int(ops.static_new())
        self._prev = []

    def declare_sorted(self, idx, n_rows):
        return ops.static_declare(idx, int(n_rows), self._owner)

    def declare_range(self, idx, hi):
        """``idx`` (any order) must lie in [0, hi): checked on the device by every ``refresh()``."""
        ops.static_declare_range(idx, int(hi), self._owner)

    def refresh(self):
        ops.static_refresh(self._owner)

    def check(self):
        f = int(ops.static_check(self._owner))
        if f:
            raise SpkHipError("StaticLists: a declared index was %s" % ("not ascending" if f & 1 else "out of range"))

    def release(self):
        """Drop this object's declarations and buffers (no graph that refreshed them may be replayed afterwards)."""
        if self._owner:
            ops.static_release(self._owner)
            self._owner = 0

    def __del__(self):
        try:
            self.release()
        except Exception:      # interpreter shutdown
            pass

    def __enter__(self):
        self._prev.append(bool(ops.static_enable(True)))
        return self

    def __exit__(self, *exc):
        ops.static_enable(self._prev.pop() if self._prev else False)      # restore, do not just switch off: scopes nest
        return False
