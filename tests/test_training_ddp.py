"""Data-parallel gradient averaging of the training path (BASELINE config 5's multi-GPU shape) on 2 gloo ranks (CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shasta_amd.training import allreduce_gradients
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 70000, 3, 1)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params.append(torch.nn.Parameter(torch.zeros(2)))  # no gradient: skipped
    allreduce_gradients(params, bucket_bytes=64 * 1024)  # forces several buckets
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params[:4]))
    ok = ok and params[4].grad is None
    # low-rank factor exchange: gathering the rank-B factors and one product over world*B rows == the all-reduced average of
    # the local outer products; a tensor flagged as already global is skipped by allreduce_gradients (and the flag is cleared)
    from shasta_amd.training import _all_gather_rows
    g = torch.Generator().manual_seed(100 + rank)
    gh, x = torch.randn(3, 5, generator=g), torch.randn(3, 7, generator=g)
    gh_all, x_all = _all_gather_rows(gh, world), _all_gather_rows(x, world)
    ok = ok and gh_all.shape == (world * 3, 5) and torch.equal(gh_all[rank * 3:(rank + 1) * 3], gh)
    low_rank = (gh_all / world).t() @ x_all
    dense = gh.t() @ x
    dist.all_reduce(dense)
    ok = ok and torch.allclose(low_rank, dense / world, atol=1e-6)
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.full((4,), float(rank))
    p._shasta_grad_is_global = True
    allreduce_gradients([p])
    ok = ok and torch.equal(p.grad, torch.full((4,), float(rank))) and p._shasta_grad_is_global is False
    q.put((rank, ok))
    dist.destroy_process_group()


def test_allreduce_gradients_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
