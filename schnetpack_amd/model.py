"""Mirror of ``NeuralNetworkPotential`` (model/base.py:132-190) -- the one caller of the hot
path -- plus helpers to assemble the benchmark models and to move batches to the device."""
from typing import Dict, Final, List, Optional

import torch
import torch.nn as nn

from . import properties
from .atomistic import Atomwise, Forces, PairwiseDistances
from .nn import CosineCutoff, GaussianRBF, BesselRBF
from .representation import PaiNN, SchNet

__all__ = ["NeuralNetworkPotential", "build_model", "batch_to_inputs"]


def _is_forces(m) -> bool:
    """The mirror's ``Forces`` or the reference's own (atomistic/response.py:14-92; same attributes) -- not a subclass that may
    have changed what the module computes."""
    if type(m) is Forces:
        return True
    t = type(m)
    return (t.__name__ == "Forces" and t.__module__ == "schnetpack.atomistic.response"
            and all(hasattr(m, a) for a in ("calc_forces", "calc_stress", "energy_key", "force_key")))


def classify_potential(model) -> int:
    """0: module-by-module.  1: the standard potential -- ``PairwiseDistances`` -> fused ``SchNet`` -> ``Atomwise`` (default
    head, summed or averaged over the molecule) -> ``Forces`` without stress: representation + head are ONE operator.
    2: ... and the only other output is Forces' -dE/dR of a summed energy: energies AND forces from the two launches.
    Works on any model with the reference's ``NeuralNetworkPotential`` layout (model/base.py:132-190), i.e. also on the
    reference's own class around the HIP modules."""
    rep, ins, outs = model.representation, list(model.input_modules), list(model.output_modules)
    is_painn = isinstance(rep, PaiNN)
    if not (isinstance(rep, (SchNet, PaiNN)) and rep._fused and len(rep.interactions) > 0):
        return 0
    if not (len(ins) == 1 and type(ins[0]) is PairwiseDistances and len(outs) >= 1):
        return 0
    head = outs[0]
    if not (isinstance(head, Atomwise) and head._fused_head and head.per_atom_output_key is None
            and head.aggregation_mode in ("sum", "avg")):
        return 0
    if not all(_is_forces(m) and not m.calc_stress for m in outs[1:]):
        return 0
    if (len(outs) == 2 and outs[1].calc_forces and outs[1].energy_key == head.output_key and head.aggregation_mode == "sum"):
        return 2
    return 0 if is_painn else 1          # (PaiNN: only the energies-and-forces form is one operator)


def potential_forward(model, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Standard potential, differentiable form: ``torch.ops.spk_hip.schnet_potential`` (pair vectors, representation and
    energy head in one launch each way); the ``Forces`` module that follows triggers the one-launch backward."""
    rep, head = model.representation, model.output_modules[0]
    idx_m = inputs[properties.idx_m]
    n_mol = head._n_molecules(inputs, idx_m)
    kind, p0, p1 = rep.radial_basis.kernel_params()
    l0, l1 = head.outnet[0], head.outnet[1]
    E, x = torch.ops.spk_hip.schnet_potential(
        rep.embed(inputs), inputs[properties.R], inputs.get(properties.offsets), inputs[properties.idx_i], inputs[properties.idx_j], idx_m,
        n_mol, rep.interaction_weights(), [l0.weight, l0.bias, l1.weight, l1.bias], rep.n_filters, kind, p0, p1,
        rep.cutoff_fn.cutoff_value(), head._head_act)
    if head.aggregation_mode == "avg":
        E = E / inputs[properties.n_atoms]
    inputs["scalar_representation"] = x
    inputs[head.output_key] = E
    return inputs


def potential_forces_forward(model, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Energies and forces straight from the two launches (no autograd node: eval only; the embedding rows are looked up
    inside the forward launch when the nuclear embedding is a plain table)."""
    rep, head, frc = model.representation, model.output_modules[0], model.output_modules[1]
    idx_m = inputs[properties.idx_m]
    kind, p0, p1 = rep.radial_basis.kernel_params()
    l0, l1 = head.outnet[0], head.outnet[1]
    plain = type(rep.embedding) is nn.Embedding and len(rep.electronic_embeddings) == 0
    if isinstance(rep, PaiNN):
        with torch.no_grad():
            E, F, x, mu = torch.ops.spk_hip.painn_potential_forces(
                None if plain else rep.embed(inputs), rep.embedding.weight if plain else None, inputs[properties.Z], inputs[properties.R],
                inputs.get(properties.offsets), inputs[properties.idx_i], inputs[properties.idx_j], idx_m, head._n_molecules(inputs, idx_m),
                rep.interaction_weights(), [l0.weight, l0.bias, l1.weight, l1.bias], rep.share_filters, rep.epsilon, kind, p0, p1,
                rep.cutoff_fn.cutoff_value(), head._head_act)
        if torch.is_grad_enabled():
            guard = [l0.weight]
            E, F = torch.ops.spk_hip.eval_guard(E, guard), torch.ops.spk_hip.eval_guard(F, guard)
        inputs["scalar_representation"] = x
        inputs["vector_representation"] = mu
        inputs[head.output_key] = E
        inputs[frc.force_key] = F
        return inputs
    with torch.no_grad():
        x0 = None if plain else rep.embed(inputs)
        E, F, x = torch.ops.spk_hip.schnet_potential_forces(
            x0, rep.embedding.weight if plain else None, inputs[properties.Z], inputs[properties.R], inputs.get(properties.offsets),
            inputs[properties.idx_i], inputs[properties.idx_j], idx_m, head._n_molecules(inputs, idx_m), rep.interaction_weights(),
            [l0.weight, l0.bias, l1.weight, l1.bias], rep.n_filters, kind, p0, p1, rep.cutoff_fn.cutoff_value(), head._head_act)
    if torch.is_grad_enabled():      # a backward pass into this eval-mode model gets the eval-only message, not silence
        guard = [l0.weight]
        E, F = torch.ops.spk_hip.eval_guard(E, guard), torch.ops.spk_hip.eval_guard(F, guard)
    inputs["scalar_representation"] = x
    inputs[head.output_key] = E
    inputs[frc.force_key] = F
    return inputs


def potential_fm_forward(model, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """TRAINING mode of the standard potential: ``torch.ops.spk_hip.schnet_fm`` / ``painn_fm`` -- energies and forces from one
    operator whose backward is the forward-over-reverse engine (csrc/spk_fm.hip): the gradient of any loss(E, F) w.r.t. every
    weight in ~100 launches, where ``Forces(create_graph=True)`` (atomistic/response.py:59-68) records a second-order graph of
    several hundred nodes.  The positions enter detached: the operator differentiates w.r.t. the weights only, the forces it
    returns ARE -dE/dR.  Consequence (documented, not detected: ``forward`` itself marks the positions as requiring grad, like the
    reference's, so a request cannot be told from the default): in train mode the outputs carry no graph to the positions --
    ``autograd.grad(E, R)`` reports an unused input; use ``model.fm_engine = False`` for position derivatives of any order."""
    rep, head, frc = model.representation, model.output_modules[0], model.output_modules[1]
    idx_m = inputs[properties.idx_m]
    kind, p0, p1 = rep.radial_basis.kernel_params()
    l0, l1 = head.outnet[0], head.outnet[1]
    common = (rep.embedding.weight, inputs[properties.Z], inputs[properties.R].detach(), inputs.get(properties.offsets), inputs[properties.idx_i],
              inputs[properties.idx_j], idx_m, head._n_molecules(inputs, idx_m), rep.interaction_weights(), [l0.weight, l0.bias, l1.weight, l1.bias])
    if isinstance(rep, PaiNN):
        E, F = torch.ops.spk_hip.painn_fm(*common, rep.share_filters, rep._eps(), kind, p0, p1, rep.cutoff_fn.cutoff_value(), model._fm_head_act)
    else:
        E, F = torch.ops.spk_hip.schnet_fm(*common, rep.n_filters, kind, p0, p1, rep.cutoff_fn.cutoff_value(), model._fm_head_act)
    inputs[head.output_key] = E
    inputs[frc.force_key] = F
    return inputs


def classify_fm(model) -> int:
    """Activation id of the energy head (> 0) when the force-matching engine covers the model, else 0: ``PairwiseDistances`` -> fused-able
    ``SchNet`` / ``PaiNN`` with a plain nuclear embedding table -> ``Atomwise`` (two Dense layers, width-1 output, summed per molecule; ANY
    hidden width) -> ``Forces`` of that energy without stress.  Looser than :func:`classify_potential`: the engine's Dense layers take any
    width, so e.g. n_atom_basis = 32 (head width 16) trains through it although the fused eval head does not cover it."""
    from . import _lib
    from .nn import Dense
    from .nn.base import activation_id
    rep, ins, outs = model.representation, list(model.input_modules), list(model.output_modules)
    if not (isinstance(rep, (SchNet, PaiNN)) and rep._fused and len(rep.interactions) > 0):
        return 0
    if not (type(rep.embedding) is nn.Embedding and len(rep.electronic_embeddings) == 0):
        return 0
    if getattr(rep.radial_basis, "trainable", False):      # the engine has no gradient w.r.t. offsets / widths: the closed operators do
        return 0
    if not (len(ins) == 1 and type(ins[0]) is PairwiseDistances and len(outs) == 2):
        return 0
    head, frc = outs
    if not (isinstance(head, Atomwise) and head.per_atom_output_key is None and head.aggregation_mode == "sum" and head.n_out == 1):
        return 0
    net = head.outnet
    if not (isinstance(net, nn.Sequential) and len(net) == 2 and all(isinstance(l, Dense) for l in net)):
        return 0
    act = activation_id(net[0].activation)
    if act not in (_lib.SPK_ACT_SSP, _lib.SPK_ACT_SILU) or activation_id(net[1].activation) != _lib.SPK_ACT_NONE:
        return 0
    if net[1].out_features != 1 or net[0].bias is None or net[1].bias is None:
        return 0
    if not (_is_forces(frc) and frc.calc_forces and not frc.calc_stress and frc.energy_key == head.output_key):
        return 0
    return int(act)


class NeuralNetworkPotential(nn.Module):
    """input_modules -> representation -> output_modules (dict in, dict out); TorchScript-able like the reference's
    (src/scripts/spkdeploy:16-40 scripts the whole model).

    The standard potential -- ``PairwiseDistances`` -> fused ``SchNet`` -> ``Atomwise`` (default head, summed or averaged over
    the molecule) -> ``Forces`` without stress -- runs in eval mode as ONE operator, ``torch.ops.spk_hip.schnet_potential``:
    on batches of small molecules the pair vectors, the representation and the energy head are one launch and the backward
    that ``Forces`` triggers (dE/dE -> head -> representation -> dE/dR) is one launch; on every other list the operator runs
    the same three stages through their own kernels.  Any other composition takes the module-by-module path below.
    (``install(fused_potential=True)`` gives the reference's own ``NeuralNetworkPotential`` the same routing.)"""

    required_derivatives: List[str]
    model_outputs: List[str]
    _potential: Final[bool]
    _potential_forces: Final[bool]
    #: training mode of the standard potential through the force-matching engine (``spk_hip::schnet_fm`` / ``painn_fm``: weight
    #: gradients of a loss(E, F) by forward-over-reverse).  Set to False for the operator-by-operator path (any-order autograd).
    fm_engine: bool
    _fm_head_act: int

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
        mode = classify_potential(self)
        self._potential = mode >= 1
        # ... and when the only other output is Forces' -dE/dR, energies AND forces come from the two launches directly
        self._potential_forces = mode == 2
        self._fm_head_act = classify_fm(self)
        self.fm_engine = self._fm_head_act > 0

    @torch.jit.unused
    def _potential_fm_forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return potential_fm_forward(self, inputs)

    @torch.jit.unused
    def _potential_forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return potential_forward(self, inputs)

    @torch.jit.unused
    def _potential_forces_forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return potential_forces_forward(self, inputs)

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for p in self.required_derivatives:
            if p in inputs:
                inputs[p].requires_grad_()
        dev_ok = properties.R in inputs and (inputs[properties.R].is_cuda or inputs[properties.R].is_meta) and inputs[properties.R].dtype == torch.float32
        if not dev_ok:           # host / non-float32 tensors: module by module, every mirror on its ATen route (nn/fallback.py)
            for m in self.input_modules:
                inputs = m(inputs)
            inputs = self.representation(inputs)
            for m in self.output_modules:
                inputs = m(inputs)
            return {k: inputs[k] for k in self.model_outputs}
        if self._potential_forces and not self.training and not torch.jit.is_scripting():
            inputs = self._potential_forces_forward(inputs)
            return {k: inputs[k] for k in self.model_outputs}
        if self.training and self.fm_engine and not torch.jit.is_scripting():
            pos = inputs[properties.R]
            w0 = self.representation.embedding.weight
            # the engine is fp32 on the device, for positions AND parameters; anything else (float64 checks of the operator-by-operator path on
            # the host, float64 weights) takes the primitives below
            if pos.is_cuda and pos.dtype == torch.float32 and w0.is_cuda and w0.dtype == torch.float32:
                inputs = self._potential_fm_forward(inputs)
                return {k: inputs[k] for k in self.model_outputs}
        if self._potential and not self.training and not torch.jit.is_scripting():
            inputs = self._potential_forward(inputs)
            for i, m in enumerate(self.output_modules):
                if i > 0:
                    inputs = m(inputs)
            return {k: inputs[k] for k in self.model_outputs}
        for m in self.input_modules:
            inputs = m(inputs)
        inputs = self.representation(inputs)
        for m in self.output_modules:
            inputs = m(inputs)
        return {k: inputs[k] for k in self.model_outputs}


def build_model(kind: str = "schnet", n_atom_basis: int = 128, n_interactions: int = 3,
                n_rbf: int = 20, cutoff: float = 5.0, radial: str = "gaussian", trainable_rbf: bool = False, **rep_kw):
    """SchNet / PaiNN + Atomwise energy head + Forces, assembled like
    configs/model/nnp.yaml:4-8 + experiment/md17.yaml:30-38."""
    rb = GaussianRBF(n_rbf, cutoff, trainable=trainable_rbf) if radial == "gaussian" else BesselRBF(n_rbf, cutoff)
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
