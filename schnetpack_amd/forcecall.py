"""Force-call graph capture.

With a fixed neighbour list the eval-mode force call (PairwiseDistances -> representation -> Atomwise
-> Forces, model/base.py:174-190) is a fixed sequence of ~25 kernel launches whose launch overhead
exceeds their run time for molecular batches.  :class:`GraphedForceCall` captures the call once per
neighbour list into a HIP graph with static input buffers and replays it; a new list (different index
tensors or shapes) triggers a re-capture.
"""
from typing import Dict, Optional

import torch

from . import properties

__all__ = ["GraphedForceCall"]


class GraphedForceCall:
    """``call(inputs) -> {"energy", "forces"}`` for an eval-mode ``NeuralNetworkPotential``.

    ``inputs`` is the reference's batch dict (``_atomic_numbers``, ``_positions``, ``_idx_i``,
    ``_idx_j``, ``_offsets``, ``_idx_m``, optionally ``_n_molecules`` / ``_n_atoms``).  Positions and
    offsets are copied into static buffers on every call; the index tensors are captured by reference
    (pass the same tensor objects while the list is unchanged, as ``NeighborListMD`` does).  The returned
    tensors are static output buffers: they are overwritten by the next call.
    """

    def __init__(self, model, energy_key: str = "energy", force_key: str = "forces", use_graph: bool = True):
        self.model = model.eval()
        self.energy_key, self.force_key = energy_key, force_key
        self.use_graph = use_graph
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._key = None
        self._static: Dict[str, torch.Tensor] = {}
        self._out = None
        self.n_captures = 0

    def _list_key(self, inputs):
        ii, jj = inputs[properties.idx_i], inputs[properties.idx_j]
        return (ii.data_ptr(), jj.data_ptr(), ii._version, jj._version, tuple(ii.shape),
                tuple(inputs[properties.R].shape), inputs[properties.Z].data_ptr())

    def _eager(self, static):
        out = self.model(dict(static))
        return out[self.energy_key].detach(), out[self.force_key].detach()

    def _capture(self, inputs):
        self._static = {k: v for k, v in inputs.items()}
        self._static[properties.R] = inputs[properties.R].detach().clone()
        if inputs.get(properties.offsets) is not None:
            self._static[properties.offsets] = inputs[properties.offsets].detach().clone()
        self._src = (inputs[properties.idx_i], inputs[properties.idx_j])   # keep the list alive
        self.graph = None
        self._out = self._eager(self._static)          # also builds the edge plan (one sync) outside the capture
        if not self.use_graph:
            return
        if self.n_captures == 0:     # allocator / lazy-init warm-up on a side stream, first capture only
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._eager(self._static)
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._out = self._eager(self._static)
        self.graph = g
        self.n_captures += 1

    def __call__(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        key = self._list_key(inputs)
        if key != self._key:
            self._capture(inputs)
            self._key = key
        with torch.no_grad():   # the model marks the static positions as requiring grad (a leaf): plain data copies
            if inputs[properties.R] is not self._static[properties.R]:
                self._static[properties.R].copy_(inputs[properties.R].detach())
            off = inputs.get(properties.offsets)
            if off is not None and off is not self._static[properties.offsets]:
                self._static[properties.offsets].copy_(off.detach())
        if self.graph is not None:
            self.graph.replay()
        else:
            self._out = self._eager(self._static)
        return {self.energy_key: self._out[0], self.force_key: self._out[1]}

    @property
    def positions(self) -> torch.Tensor:
        """The static position buffer; integrators may update it in place and then call
        ``replay()`` to avoid the copy."""
        return self._static[properties.R]

    def replay(self) -> Dict[str, torch.Tensor]:
        if self.graph is not None:
            self.graph.replay()
        else:
            self._out = self._eager(self._static)
        return {self.energy_key: self._out[0], self.force_key: self._out[1]}
