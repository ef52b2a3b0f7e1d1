"""Static-shape training step replayed as HIP graphs (SURVEY.md section 8, configs[3] / cfg 4).

The differentiable path (``module.training``: HIP scatter/gather/Dense primitives, double backward through
``Forces(create_graph=True)``) is a few hundred small launches per step at rMD17 batch sizes -- launch bound.
Instead of a tracing compiler the step is captured once into a HIP graph and replayed:

* shapes are made static by padding the neighbour list of every batch to ``max_edges`` with inert pairs
  (self pairs of the last atom with an offset beyond the cutoff: the cosine cutoff and its derivative vanish
  there, so energies, forces and all gradients are exactly those of the unpadded batch);
* the index tensors are static BUFFERS refilled per step; their CSR row pointers are recomputed inside the
  graph by a device-only kernel (``torchops.StaticLists``), no plan cache, no host round trip;
* gradients leave the backward as the tensors its last kernels wrote (no accumulate-into-bucket launch per parameter),
  are gathered into one flat bucket by ONE concatenation (``parallel.FlatGradAllReduce.pack``) and re-bound as views of
  it; the bucket is all-reduced with one RCCL call between the two graphs (backward | optimizer) when there are ranks;
* AdamW runs over the flat bucket as ONE launch of this library (:class:`FlatAdamW`; ``optimizer="torch"``: the framework's
  ``capturable`` + ``fused`` multi-tensor form), its step count on the device, so that it is one node of the graph.

The first two calls of :meth:`GraphedTrainStep.step` run eagerly (allocator / autotune warm-up; they are
real optimizer steps), the third captures, every later one replays.
"""
from typing import Dict, Optional

import ctypes

import torch

from . import properties, torchops
from .parallel import FlatGradAllReduce

__all__ = ["GraphedTrainStep", "FlatAdamW", "pad_edges"]


def pad_edges(idx_i, idx_j, offsets, n_atoms: int, max_edges: int, cutoff: float):
    """Pad a neighbour list (idx_i ascending) to ``max_edges`` pairs with inert entries
    (i = j = n_atoms - 1, offset = (2 cutoff + 1, 0, 0)): d = 2 cutoff + 1 > cutoff => f_cut = f_cut' = 0."""
    E = int(idx_i.shape[0])
    if E > max_edges:
        raise ValueError("neighbour list has %d pairs, capacity is %d" % (E, max_edges))
    pad = max_edges - E
    last = n_atoms - 1
    ii = torch.cat([idx_i, torch.full((pad,), last, dtype=idx_i.dtype, device=idx_i.device)])
    jj = torch.cat([idx_j, torch.full((pad,), last, dtype=idx_j.dtype, device=idx_j.device)])
    po = torch.zeros((pad, 3), dtype=offsets.dtype, device=offsets.device)
    po[:, 0] = 2.0 * cutoff + 1.0
    return ii, jj, torch.cat([offsets, po])


class FlatAdamW:
    """``torch.optim.AdamW`` arithmetic (the optimizer of the reference's training configs, task.py:253-275) as ONE launch for all
    parameters: the gradients are the flat bucket of a :class:`FlatGradAllReduce` (``reducer.flat``), the moments are flat buffers of the
    same layout, the parameters stay the model's own tensors (``spk_adamw_devlr_f32`` reaches them through a chunk table).  The step count
    AND the learning rate live on the device, so a captured step replays and follows a schedule (``opt.lr = ...`` between replays is one
    4-byte copy).  State: ``exp_avg``, ``exp_avg_sq``, ``step_count``, ``lr`` (:meth:`state_dict` / :meth:`load_state_dict`).

    Differences from ``torch.optim.AdamW``, all deliberate: one parameter group; no ``amsgrad`` / ``maximize``; a parameter the backward
    never reached is treated as a zero gradient (torch skips it, so its weight decay is skipped there too); the parameters are written
    through raw pointers, so their autograd version counters do not move -- every eager :meth:`step` therefore tells the operator
    library that its cached weight images are stale (``spk_hip::weights_changed``; a captured step cannot, the owner of the graph
    does it after the replay, see :meth:`GraphedTrainStep.step`); the float32 step count saturates at 2**24, where both bias corrections
    have long been 1."""

    CHUNK = 2048

    def __init__(self, reducer, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        from . import _lib
        self._lib = _lib
        self.reducer = reducer
        self.betas, self.eps, self.weight_decay = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        params = reducer.params
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        rows, off = [], 0
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FlatAdamW: contiguous float32 parameters on one device")
            n = p.numel()
            for c0 in range(0, n, self.CHUNK):
                rows.append((p.data_ptr(), c0, off + c0, min(self.CHUNK, n - c0)))
            off += n
        self.numel = off
        self._params = list(params)            # (keeps the storages the table points into alive)
        self._ptrs = [p.data_ptr() for p in self._params]
        self.chunks = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        self.step_count = torch.zeros(1, device=dev)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr = float(lr)
        self._lr_dev = torch.full((1,), float(lr), device=dev)
        # the face a learning-rate schedule needs: param_groups[0]["lr"] + sync_lr().  torch.optim.lr_scheduler classes check isinstance(optimizer,
        # torch.optim.Optimizer) and cannot be attached to this object: compute the schedule externally (or drive a scheduler on a throw-away torch
        # optimizer with the same initial lr) and write `opt.lr = value` / `opt.param_groups[0]["lr"] = value; opt.sync_lr()` once per step
        self.param_groups = [{"params": self._params, "lr": float(lr), "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}]

    # ------------------------------------------------------------ learning rate (host mirror + device scalar)
    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):
        value = float(value)
        if value < 0.0:
            raise ValueError("FlatAdamW: negative learning rate")
        if value != self._lr:
            self._lr = value
            self._lr_dev.fill_(value)          # outside any capture: the next replay reads the new value
        self.param_groups[0]["lr"] = value

    def sync_lr(self):
        """Adopt ``param_groups[0]["lr"]`` (what a torch-style scheduler wrote)."""
        self.lr = self.param_groups[0]["lr"]

    # ------------------------------------------------------------ checkpointing (task checkpoints carry the optimizer state)
    def state_dict(self):
        return {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(), "step_count": self.step_count.detach().clone(),
                "lr": self._lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "numel": self.numel}

    def load_state_dict(self, state):
        """Restore a :meth:`state_dict` (a flat layout -- not interchangeable with ``torch.optim.AdamW``'s per-parameter state).  Everything is
        validated BEFORE anything is copied: a mismatch leaves the optimizer untouched."""
        if int(state["numel"]) != self.numel:
            raise ValueError("FlatAdamW: state of %d elements, optimizer of %d" % (int(state["numel"]), self.numel))
        # betas / eps / weight_decay are launch arguments baked into a captured graph: a different value needs a re-capture, so it is refused here
        for k, have in (("betas", self.betas), ("eps", self.eps), ("weight_decay", self.weight_decay)):
            want = tuple(float(v) for v in state[k]) if k == "betas" else float(state[k])
            if want != have:
                raise ValueError("FlatAdamW.load_state_dict: %s = %r differs from this optimizer's %r" % (k, want, have))
        for k in ("exp_avg", "exp_avg_sq"):
            if state[k].numel() != self.numel:
                raise ValueError("FlatAdamW.load_state_dict: %s has %d elements, optimizer %d" % (k, state[k].numel(), self.numel))
        with torch.no_grad():
            self.exp_avg.copy_(state["exp_avg"])
            self.exp_avg_sq.copy_(state["exp_avg_sq"])
            self.step_count.copy_(state["step_count"])
        self.lr = state["lr"]

    def zero_grad(self, set_to_none: bool = True):
        """The gradients are the views of the reducer's bucket: released (the next backward re-creates them)."""
        self.reducer.release()

    def step(self):
        flat = self.reducer.flat
        if flat is None or flat.numel() != self.numel:
            raise RuntimeError("FlatAdamW: the gradient bucket is not bound (run the backward and reducer.pack() first)")
        if any(p.data_ptr() != q for p, q in zip(self._params, self._ptrs)):
            raise RuntimeError("FlatAdamW: a parameter was re-allocated after the optimizer was built")
        L, c = self._lib, ctypes.c_void_p
        L.check(L.lib().spk_adamw_devlr_f32(c(self.chunks.data_ptr()), int(self.chunks.shape[0]), L.fptr(flat), L.fptr(self.exp_avg), L.fptr(self.exp_avg_sq),
                                            L.fptr(self.step_count), c(self._ticket.data_ptr()), L.fptr(self._lr_dev), self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, L.stream()))
        # the parameters changed behind torch's back (no version bump): every cached transposed / packed weight image of the operator library and
        # every filter table is stale now.  (Inside a capture the call would only run once, at capture time: the graph's owner repeats it per replay.)
        if not torch.cuda.is_current_stream_capturing():
            torch.ops.spk_hip.weights_changed()


class GraphedTrainStep:
    """Force-matching training step ``loss = w_E MSE(E) + w_F MSE(F)`` (the reference's task block,
    examples/.../config.yaml) for batches of a fixed number of atoms / molecules and at most ``max_edges``
    pairs.  ``load(batch, E_target, F_target)`` fills the static buffers, ``step()`` runs one optimizer step
    and returns the (device) loss of that step."""

    def __init__(self, model, n_atoms: int, n_molecules: int, max_edges: int, cutoff: float, lr: float = 1e-3,
                 loss_weights=(0.01, 0.99), group=None, use_graph: bool = True, warmup_steps: int = 2, optimizer: str = "flat"):
        self.model = model.train()
        dev = next(model.parameters()).device
        self.dev, self.N, self.M, self.Emax, self.cutoff = dev, int(n_atoms), int(n_molecules), int(max_edges), float(cutoff)
        self.wE, self.wF = loss_weights
        self.group, self.use_graph, self.warmup_steps = group, use_graph, warmup_steps
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        # the static inputs of the step are views of TWO buffers (all indices | all floats): a batch that arrives packed
        # (:meth:`pack`, e.g. from a DataLoader worker) is loaded with two copies instead of eleven
        N, Em, M = self.N, self.Emax, self.M
        self._ibuf = z(2 * N + 2 * Em, dt=torch.int64)
        self._fbuf = z(3 * N + 3 * Em + M + 3 * N)
        self.buf = {
            properties.Z: self._ibuf[:N], properties.idx_m: self._ibuf[N:2 * N],
            properties.idx_i: self._ibuf[2 * N:2 * N + Em], properties.idx_j: self._ibuf[2 * N + Em:],
            properties.R: self._fbuf[:3 * N].view(N, 3), properties.offsets: self._fbuf[3 * N:3 * N + 3 * Em].view(Em, 3),
        }
        self.E_t = self._fbuf[3 * N + 3 * Em:3 * N + 3 * Em + M]
        self.F_t = self._fbuf[3 * N + 3 * Em + M:].view(N, 3)
        self.loss = z(())
        self.lists = torchops.StaticLists()
        self.lists.declare_sorted(self.buf[properties.idx_i], self.N)
        self.lists.declare_sorted(self.buf[properties.idx_m], self.M)
        self.lists.declare_range(self.buf[properties.idx_j], self.N)
        n_emb = getattr(getattr(model.representation, "embedding", None), "num_embeddings", None)
        if n_emb is not None:
            self.lists.declare_range(self.buf[properties.Z], int(n_emb))
        self._pad_offsets = torch.zeros(self.Emax, 3, device=dev)
        self._pad_offsets[:, 0] = 2.0 * self.cutoff + 1.0
        self.reducer = FlatGradAllReduce(model.parameters(), as_views=True)
        # "flat" (default): AdamW over the flat gradient bucket in one launch of this library (FlatAdamW); "torch": torch.optim.AdamW
        # capturable + fused -- one multi-tensor kernel of the framework plus a launch for the step counters (13 / 31 us per step for the
        # 30 / 34 tensors of SchNet / PaiNN against ~3 us; the foreach form costs two broadcast divisions PER PARAMETER on top)
        if optimizer == "flat" and dev.type == "cuda":
            self.opt = FlatAdamW(self.reducer, lr=lr)
        elif optimizer in ("flat", "torch"):
            self.opt = torch.optim.AdamW(model.parameters(), lr=lr, capturable=True, fused=True)
        else:
            raise ValueError("optimizer: 'flat' or 'torch'")
        self.n_steps = 0
        self.g_bwd = self.g_opt = None

    # ---------------------------------------------------------------- data
    def load(self, batch: Dict[str, torch.Tensor], E_target: torch.Tensor, F_target: torch.Tensor):
        """batch: Z, R, idx_i, idx_j, offsets, idx_m (host or device tensors of the synthetic / collate layout).  The pair
        list is written into the static buffers and padded there (see :func:`pad_edges`); index validation runs on the
        device inside the step (``StaticLists.refresh``: idx_i / idx_m ascending and in range, idx_j and Z in range) and is
        polled by :meth:`check` -- the kernels themselves never read or scatter out of bounds."""
        if int(batch["Z"].shape[0]) != self.N:
            raise ValueError("batch has %d atoms, the step was built for %d" % (batch["Z"].shape[0], self.N))
        E = int(batch["idx_i"].shape[0])
        if E > self.Emax:
            raise ValueError("neighbour list has %d pairs, capacity is %d" % (E, self.Emax))
        b = self.buf
        with torch.no_grad():
            b[properties.Z].copy_(batch["Z"], non_blocking=True)
            b[properties.R].copy_(batch["R"], non_blocking=True)
            b[properties.idx_m].copy_(batch["idx_m"], non_blocking=True)
            b[properties.idx_i][:E].copy_(batch["idx_i"], non_blocking=True)
            b[properties.idx_j][:E].copy_(batch["idx_j"], non_blocking=True)
            b[properties.offsets][:E].copy_(batch["offsets"], non_blocking=True)
            if E < self.Emax:                                   # inert tail: self pairs of the last atom beyond the cutoff
                b[properties.idx_i][E:].fill_(self.N - 1)
                b[properties.idx_j][E:].fill_(self.N - 1)
                b[properties.offsets][E:].copy_(self._pad_offsets[E:])
            self.E_t.copy_(E_target, non_blocking=True)
            self.F_t.copy_(F_target, non_blocking=True)

    def pack(self, batch: Dict[str, torch.Tensor], E_target: torch.Tensor, F_target: torch.Tensor, device=None):
        """The batch in the layout of the two static buffers -- (indices [2 N + 2 E_max] int64, floats [6 N + 3 E_max + M]) -- with the pair
        list already padded to the capacity (:func:`pad_edges`).  Host-side work (collate / DataLoader worker); :meth:`load_packed` then
        moves a batch with two copies."""
        if int(batch["Z"].shape[0]) != self.N:
            raise ValueError("batch has %d atoms, the step was built for %d" % (batch["Z"].shape[0], self.N))
        cpu = lambda t, dt: t.detach().to("cpu", dt)
        ii, jj, off = pad_edges(cpu(batch["idx_i"], torch.int64), cpu(batch["idx_j"], torch.int64), cpu(batch["offsets"], torch.float32), self.N,
                                self.Emax, self.cutoff)
        ip = torch.cat([cpu(batch["Z"], torch.int64), cpu(batch["idx_m"], torch.int64), ii, jj])
        fp = torch.cat([cpu(batch["R"], torch.float32).reshape(-1), off.reshape(-1), cpu(E_target, torch.float32).reshape(-1),
                        cpu(F_target, torch.float32).reshape(-1)])
        if ip.numel() != self._ibuf.numel() or fp.numel() != self._fbuf.numel():
            raise ValueError("packed batch does not match the static buffers (n_molecules / capacity)")
        if device is not None:
            return ip.to(device), fp.to(device)
        return ip, fp

    def load_packed(self, ipack: torch.Tensor, fpack: torch.Tensor):
        with torch.no_grad():
            self._ibuf.copy_(ipack, non_blocking=True)
            self._fbuf.copy_(fpack, non_blocking=True)

    # ---------------------------------------------------------------- the step
    def _forward_backward(self):
        self.lists.refresh()
        self.reducer.release()
        inputs = dict(self.buf)
        inputs[properties.R] = self.buf[properties.R].detach().requires_grad_(False)
        inputs["_n_molecules"] = self.M
        out = self.model(inputs)
        # w_E MSE(E) + w_F MSE(F) and both gradients as ONE launch (the framework arithmetic of the two terms and their backward: 19)
        loss = torch.ops.spk_hip.fm_loss(out["energy"], self.E_t, out["forces"], self.F_t, float(self.wE), float(self.wF))
        loss.backward()
        self.reducer.pack()
        self.loss.copy_(loss.detach())

    def _eager_step(self):
        with self.lists:
            self._forward_backward()
        self.reducer(self.group)
        self.opt.step()

    def _capture(self):
        torch.cuda.synchronize(self.dev)
        multi = self.group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                           and torch.distributed.get_world_size() > 1)
        self.g_bwd = torch.cuda.CUDAGraph()
        with self.lists:
            with torch.cuda.graph(self.g_bwd):
                self._forward_backward()
                if not multi:
                    self.opt.step()
        if multi:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_bwd.pool()):
                self.opt.step()

    def step(self) -> torch.Tensor:
        self.n_steps += 1
        if not self.use_graph or self.n_steps <= self.warmup_steps:
            self._eager_step()
            return self.loss
        if self.g_bwd is None:
            self._capture()
        self.g_bwd.replay()
        if self.g_opt is not None:
            self.reducer(self.group)
            self.g_opt.replay()
        # the captured optimizer updates the parameters in place WITHOUT moving their version counters: tell the operator
        # library that every transposed / packed weight copy it caches (eval-mode forwards, e.g. validation) is stale
        torch.ops.spk_hip.weights_changed()
        return self.loss

    def check(self):
        """Poll the device-side validity flag of the declared index tensors (one D2H)."""
        self.lists.check()
