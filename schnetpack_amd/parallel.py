"""Multi-GPU: one process per GPU; the path shards by independent molecules / MD replicas
(SURVEY.md section 8(e)).  No collective inside the force call.  Training adds exactly one
all-reduce of one flat gradient bucket per step (RCCL over xGMI: 2.36 MB for PaiNN(128,3) is
latency-bound, so a single bucket beats per-parameter hooks).  Works on any torch.distributed
backend (``nccl`` = RCCL on ROCm; ``gloo`` in the CPU tests)."""
import os
from typing import Iterable, List, Tuple

import torch


def shard_frames(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk [lo, hi) of ``n_total`` independent systems owned by ``rank``; sizes differ
    by at most one and cover the range exactly."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_systems(systems: List, rank: int, world: int) -> List:
    lo, hi = shard_frames(len(systems), rank, world)
    return systems[lo:hi]


class FlatGradAllReduce:
    """Average the gradients of ``params`` over all ranks with ONE all-reduce of one flat fp32
    buffer (what DDP does with a single bucket).

    ``as_views=False``: gradients are copied into the bucket and back; parameters without a gradient
    contribute zeros so that every rank reduces the same layout.
    ``as_views=True`` (DDP's ``gradient_as_bucket_view``): every ``p.grad`` IS a view of the bucket, so
    autograd accumulates straight into it and the step costs no copy kernels; clear the gradients
    with :meth:`zero` (``optimizer.zero_grad(set_to_none=True)`` would drop the views)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], as_views: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.as_views = as_views
        self.collectives = 0            # all-reduces issued so far (see _reduce)
        self.last_reduced_ptr = None
        self.last_reduced_numel = 0
        if as_views and self.params:
            self._bind()

    def _bind(self):
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        """Clear all gradients with one fill (views mode)."""
        if self.flat is not None:
            self.flat.zero_()

    def release(self):
        """Drop the gradients (``p.grad = None``) so that the next backward hands its tensors over instead of adding them
        into the bucket views -- one launch per parameter saved; follow the backward with :meth:`pack`."""
        for p in self.params:
            p.grad = None

    def pack(self):
        """Gather the freshly produced gradients into the flat bucket with ONE concatenation and re-bind every ``p.grad`` as
        a view of it (so the all-reduce and the optimizer see one buffer).  Parameters the backward did not reach count as
        zeros."""
        if not self.params:
            return
        if self.flat is None:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.params[0].device)
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        torch.cat(parts, out=self.flat)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def _reduce(self, group):
        """The one collective of a training step.  Issued when the job has more than one rank -- and also at world size 1 when
        the caller names a process ``group`` or ``SPK_FORCE_COLLECTIVES=1`` is set (as ``SPK_MD_FORCE_COLLECTIVES`` does for the
        bead exchange, md.py): the RCCL path of a one-GPU box is then the path an 8-GPU job takes, and a test can count it.
        ``self.collectives`` counts the all-reduces issued, ``self.last_reduced_ptr`` / ``last_reduced_numel`` say which buffer
        went to the backend (the flat bucket itself: no staging copy)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world > 1 or group is not None or os.environ.get("SPK_FORCE_COLLECTIVES", "0") not in ("", "0"):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.collectives += 1
            self.last_reduced_ptr = self.flat.data_ptr()
            self.last_reduced_numel = self.flat.numel()
            if world > 1:
                self.flat.div_(world)

    def __call__(self, group=None):
        if not self.params:
            return
        if self.as_views:
            off = 0
            for p in self.params:     # a gradient that was re-created (e.g. after set_to_none) is re-bound
                n = p.numel()
                view = self.flat[off:off + n]
                if p.grad is None:
                    view.zero_()
                    p.grad = view.view_as(p)
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad.reshape(-1))
                    p.grad = view.view_as(p)
                off += n
            self._reduce(group)
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self._reduce(group)
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def gather_sharded_results(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank results (e.g. energies of the rank's frames) in rank order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    mx = int(max(int(s) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[: int(s)] for o, s in zip(outs, sizes)], 0)
