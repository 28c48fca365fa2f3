"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel plumbing in fusiondepth_amd/dp.py: flat
parameter/gradient buffers, bucketed backward-overlapped all-reduce, unused-parameter handling, broadcast.
The HIP kernels are not involved (no GPU here): the "model" is a tiny torch module and the update is plain SGD."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.unused = torch.nn.Linear(16, 1000)      # like the ResNet `fc`: in parameters(), never gets a gradient

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, out):
    from fusiondepth_amd import dp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                     # different init per rank: broadcast must fix it
    net = Net()
    dp.broadcast_module_state([net])
    flat = dp.FlatParameters(list(net.parameters()))
    sync = dp.GradientSynchronizer(flat, world, bucket_bytes=256)     # tiny buckets => several all-reduces
    assert len(sync.buckets) >= 3
    torch.manual_seed(7)
    data = torch.randn(2 * world, 2, 3, 8)           # [rank slot, micro-batch, sample, feature]
    for step in range(3):
        for mb in range(2):                           # accumulate 2 micro-batches, reduce on the last one
            x = data[rank * 2 + (step % 2) * 0, mb]
            if mb == 1:
                sync.arm()
            (net(x).pow(2).mean() / 2).backward()
        scale = sync.finish()
        with torch.no_grad():
            flat.flat_param -= 0.1 * scale * flat.flat_grad
        flat.zero_grad()
    out[rank] = flat.flat_param.clone()
    # reference: single process, same init (rank 0's), global batch = concat of the ranks' data
    if rank == 0:
        torch.manual_seed(100)
        ref = Net()
        for step in range(3):
            ref.zero_grad()
            loss = 0
            for rr in range(world):
                for mb in range(2):
                    loss = loss + ref(data[rr * 2, mb]).pow(2).mean() / 2
            (loss / world).backward()
            with torch.no_grad():
                for p in ref.parameters():
                    if p.grad is not None:
                        p -= 0.1 * p.grad
        out["ref"] = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1]), "replicas diverged"
    assert torch.allclose(out[0], out["ref"], rtol=1e-5, atol=1e-6), float((out[0] - out["ref"]).abs().max())


def test_flat_parameters_views_and_buckets():
    from fusiondepth_amd import dp
    net = Net()
    before = [p.detach().clone() for p in net.parameters()]
    flat = dp.FlatParameters(list(net.parameters()))
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b)
        assert p.data_ptr() >= flat.flat_param.data_ptr()
    net(torch.randn(3, 8)).sum().backward()
    assert flat.flat_grad.abs().sum() > 0                # autograd accumulated straight into the flat buffer
    n_used = sum(p.numel() for n, p in net.named_parameters() if not n.startswith("unused"))
    assert flat.flat_grad[n_used:].abs().sum() == 0       # the unused head received nothing
    flat.zero_grad()
    assert flat.flat_grad.abs().sum() == 0
    sync = dp.GradientSynchronizer(flat, 1)
    sync.arm()
    assert sync.finish() == 1.0


class _InPlaceLinear(torch.autograd.Function):
    """Stand-in for the HIP backward kernels: accumulates the weight gradient straight into ``w.grad`` (a view of the flat
    buffer), returns no parameter gradient to autograd and reports completion through functional._grad_ready."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        from fusiondepth_amd import functional as FD
        (x,) = ctx.saved_tensors
        ctx.w.grad += gy.t() @ x
        FD._grad_ready(ctx.w)
        return gy @ ctx.w, None


def _worker_direct(rank, world, port, out):
    from fusiondepth_amd import dp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env(backend="gloo")
    torch.manual_seed(5)
    ws = [torch.nn.Parameter(torch.randn(16, 8) * 0.3), torch.nn.Parameter(torch.randn(16, 16) * 0.3),
          torch.nn.Parameter(torch.randn(4, 16) * 0.3), torch.nn.Parameter(torch.randn(7, 3))]       # the last one is never used
    flat = dp.FlatParameters(ws)
    sync = dp.GradientSynchronizer(flat, world, bucket_bytes=1 << 30, segments=[1, 2, 1])   # segment boundaries cut the buckets
    assert [b[2] for b in sync.buckets] == [1, 2, 1]
    x = torch.randn(world, 5, 8, generator=torch.Generator().manual_seed(9))[rank]
    sync.arm()
    h = x
    for w in ws[:3]:
        h = torch.tanh(_InPlaceLinear.apply(h, w))
    h.pow(2).mean().backward()
    n_over = sync.n_overlapped
    scale = sync.finish()
    out[rank] = (flat.flat_grad.clone() * scale, n_over)
    if rank == 0:                                           # single process over the concatenated shards
        ref = [w.detach().clone().requires_grad_(True) for w in ws[:3]]
        xs = torch.randn(world, 5, 8, generator=torch.Generator().manual_seed(9))
        tot = 0
        for r in range(world):
            h = xs[r]
            for w in ref:
                h = torch.tanh(h @ w.t())
            tot = tot + h.pow(2).mean()
        (tot / world).backward()
        out["ref"] = torch.cat([w.grad.reshape(-1) for w in ref])
    dist.barrier()
    dist.destroy_process_group()


def test_in_place_gradients_trigger_overlapped_buckets():
    """The trainer's kernels accumulate parameter gradients in place, so autograd's hooks never fire: the buckets must leave
    through functional.add_grad_ready_callback instead, one per segment, and the unused parameter's bucket at finish()."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_direct, args=(world, _free_port(), out), nprocs=world, join=True)
    (g0, n0), (g1, n1) = out[0], out[1]
    assert n0 == n1 == 2                                     # segments 1 and 2 went out during backward, the unused one at finish
    assert torch.equal(g0, g1)
    n = out["ref"].numel()
    assert torch.allclose(g0[:n], out["ref"], rtol=1e-5, atol=1e-7) and float(g0[n:].abs().max()) == 0.0
