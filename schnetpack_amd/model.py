"""Mirror of ``NeuralNetworkPotential`` (model/base.py:132-190) -- the one caller of the hot
path -- plus helpers to assemble the benchmark models and to move batches to the device."""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import properties
from .atomistic import Atomwise, Forces, PairwiseDistances
from .nn import CosineCutoff, GaussianRBF, BesselRBF
from .representation import PaiNN, SchNet

__all__ = ["NeuralNetworkPotential", "build_model", "batch_to_inputs"]


class NeuralNetworkPotential(nn.Module):
    """input_modules -> representation -> output_modules (dict in, dict out); TorchScript-able like the reference's
    (src/scripts/spkdeploy:16-40 scripts the whole model)."""

    required_derivatives: List[str]
    model_outputs: List[str]

    def __init__(self, representation: nn.Module, input_modules: List[nn.Module] = None,
                 output_modules: List[nn.Module] = None):
        super().__init__()
        self.representation = representation
        self.input_modules = nn.ModuleList(input_modules)
        self.output_modules = nn.ModuleList(output_modules)
        self.required_derivatives = []
        for m in self.modules():
            for p in getattr(m, "required_derivatives", None) or []:
                if p not in self.required_derivatives:
                    self.required_derivatives.append(p)
        outs = []
        for m in self.modules():
            for k in getattr(m, "model_outputs", None) or []:
                if k not in outs:
                    outs.append(k)
        self.model_outputs = outs

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for p in self.required_derivatives:
            if p in inputs:
                inputs[p].requires_grad_()
        for m in self.input_modules:
            inputs = m(inputs)
        inputs = self.representation(inputs)
        for m in self.output_modules:
            inputs = m(inputs)
        return {k: inputs[k] for k in self.model_outputs}


def build_model(kind: str = "schnet", n_atom_basis: int = 128, n_interactions: int = 3,
                n_rbf: int = 20, cutoff: float = 5.0, radial: str = "gaussian", **rep_kw):
    """SchNet / PaiNN + Atomwise energy head + Forces, assembled like
    configs/model/nnp.yaml:4-8 + experiment/md17.yaml:30-38."""
    rb = GaussianRBF(n_rbf, cutoff) if radial == "gaussian" else BesselRBF(n_rbf, cutoff)
    cf = CosineCutoff(cutoff)
    if kind == "schnet":
        rep = SchNet(n_atom_basis, n_interactions, rb, cf, **rep_kw)
    elif kind == "painn":
        rep = PaiNN(n_atom_basis, n_interactions, rb, cf, **rep_kw)
    else:
        raise ValueError(kind)
    return NeuralNetworkPotential(rep, input_modules=[PairwiseDistances()],
                                  output_modules=[Atomwise(n_in=n_atom_basis, output_key=properties.energy),
                                                  Forces()])


def load_reference_params(model: NeuralNetworkPotential, rep_params, head_params):
    """Load parameters given with the reference's state_dict key names (representation keys and
    ``outnet.*`` head keys)."""
    sd = model.representation.state_dict()
    missing = [k for k in sd if k not in rep_params]
    if missing:
        raise KeyError("missing representation parameters: %s" % missing)
    model.representation.load_state_dict({k: rep_params[k].to(sd[k].dtype) for k in sd})
    model.output_modules[0].load_state_dict({k: v for k, v in head_params.items()})
    return model


def batch_to_inputs(batch, device) -> Dict[str, torch.Tensor]:
    """Synthetic batch (schnetpack_amd.synthetic) -> the reference's input dict on ``device``."""
    inp = {
        properties.Z: batch["Z"].to(device),
        properties.R: batch["R"].to(device).float().clone(),
        properties.idx_i: batch["idx_i"].to(device),
        properties.idx_j: batch["idx_j"].to(device),
        properties.offsets: batch["offsets"].to(device).float(),
        properties.idx_m: batch["idx_m"].to(device),
        "_n_molecules": int(batch["n_mol"]),
    }
    return inp
