"""Plain-ATen route for tensors the HIP kernels do not take ("unsupported combos are not errors -- the Python
module routes to the torch fallback").

The kernels are fp32 on a ROCm device.  The reference's L0 functions run on any device and dtype (nn/scatter.py:7-34,
nn/base.py:52-55, nn/radial.py:11-14 / :105-109, nn/cutoff.py:30-32); MD precision is a config switch (md/md_configs/config.yaml:4,
md/cli.py:326-327: ``simulator.to(device).to(precision)``), and BASELINE configs[0] is a force evaluation on the host.  Every module
mirror therefore asks :func:`use_aten` about its input: a host tensor or a non-float32 tensor takes the reference's own formula,
written here / in the mirrors with ATen calls (own code -- nothing under ``oracle/`` is imported by the product), differentiable to
any order through autograd like the reference.  It is a compatibility route, never a timed one: one ``warnings.warn`` per process
says so.  The raw operators (``torch.ops.spk_hip.*``) and the C ABI keep refusing such tensors.
"""
import warnings

import torch

__all__ = ["use_aten", "note_fallback", "cosine_cutoff_aten", "gaussian_rbf_aten", "bessel_rbf_aten"]

_NOTED = False


def use_aten(x: torch.Tensor) -> bool:
    """True when ``x`` lies outside what the HIP kernels cover (host memory, or any dtype but float32).  Meta tensors go to the
    operators (their Meta kernels infer the shapes of a whole force call on the build box)."""
    return (not (x.is_cuda or x.is_meta)) or x.dtype != torch.float32


@torch.jit.unused
def _note() -> None:
    global _NOTED
    if not _NOTED:
        _NOTED = True
        warnings.warn("schnetpack_amd: a host / non-float32 tensor takes the plain ATen route of the module mirrors (the reference's formulas); "
                      "the HIP kernels run float32 tensors on the ROCm device", stacklevel=3)


def note_fallback() -> None:
    """One warning per process (skipped inside TorchScript)."""
    if not torch.jit.is_scripting():
        _note()
