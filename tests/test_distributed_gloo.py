"""World-size-2 CPU (gloo) tests of the N>1 host path: frame sharding covers the batch exactly
and the single-bucket gradient all-reduce averages like DDP."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from schnetpack_amd.parallel import FlatGradAllReduce, gather_sharded_results, shard_frames
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.zeros(4))  # never receives a gradient on rank 1
    params = list(lin.parameters()) + [extra]
    x = torch.arange(10.0).view(2, 5) * (rank + 1)
    loss = lin(x).sum() + (extra.sum() if rank == 0 else 0.0)
    loss.backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    FlatGradAllReduce(params)()
    lo, hi = shard_frames(7, rank, world)
    mine = torch.arange(lo, hi, dtype=torch.float32)
    allv = gather_sharded_results(mine)
    q.put((rank, [g.tolist() for g in local], [p.grad.tolist() for p in params], allv.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, local, reduced, allv = q.get(timeout=120)
        res[rank] = (local, reduced, allv)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = torch.tensor
    for k in range(3):
        mean = (T(res[0][0][k]) + T(res[1][0][k])) / 2
        assert torch.allclose(T(res[0][1][k]), mean) and torch.allclose(T(res[1][1][k]), mean)
    assert res[0][2] == list(range(7)) and res[1][2] == list(range(7))
