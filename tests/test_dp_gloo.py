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


# ---- world size 8: the layout the driver's 8-GPU run will use ------------------------------------------
class _Branch(torch.nn.Module):
    """One 'network' of the trainer in miniature: two layers whose weight gradients are accumulated in place (like the HIP
    kernels do), plus a head that is in parameters() but never used - the ResNet ``fc`` of every encoder."""

    def __init__(self, d_in, width, with_fc):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(width, d_in) * 0.3)
        self.w2 = torch.nn.Parameter(torch.randn(d_in, width) * 0.3)
        if with_fc:
            self.fc_w = torch.nn.Parameter(torch.randn(10, width))
            self.fc_b = torch.nn.Parameter(torch.randn(10))

    def forward(self, x):
        return torch.tanh(_InPlaceLinear.apply(torch.tanh(_InPlaceLinear.apply(x, self.w1)), self.w2))


def _branches():
    # four "encoders" (with an unused head) and two "decoders", different sizes so that bucket boundaries fall unevenly
    return torch.nn.ModuleList([_Branch(8, 24, True), _Branch(8, 40, True), _Branch(8, 16, False), _Branch(8, 56, True),
                                _Branch(8, 32, True), _Branch(8, 12, False)])


def _worker8(rank, world, port, out):
    from fusiondepth_amd import dp, functional as FD
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env(backend="gloo")
    torch.set_num_threads(1)
    torch.manual_seed(300 + rank)                         # every rank starts elsewhere: the broadcast must make them rank 0's
    nets = _branches()
    dp.broadcast_module_state(nets)
    params = [p for n in nets for p in n.parameters()]
    unused = [p for n in nets for name, p in n.named_parameters() if name.startswith("fc_")]
    flat = dp.FlatParameters(params)
    sync = dp.GradientSynchronizer(flat, world, bucket_bytes=1500, segments=[len(list(n.parameters())) for n in nets], never_used=unused)
    data = torch.randn(world, 3, 5, 8, generator=torch.Generator().manual_seed(11))      # [rank, step, sample, feature]
    overl = []
    for step in range(3):
        FD.begin_forward_pass()
        sync.arm()
        loss = 0
        for n in nets:
            loss = loss + n(data[rank, step]).pow(2).mean()
        loss.backward()
        overl.append(sync.n_overlapped)
        scale = sync.finish()
        assert scale == 1.0 / world
        with torch.no_grad():
            flat.flat_param -= 0.05 * scale * flat.flat_grad
        flat.zero_grad()
    out[rank] = (flat.flat_param.clone(), overl, [list(b) for b in sync.buckets])
    if rank == 0:
        torch.manual_seed(300)
        ref = _branches()
        rp = [p for n in ref for p in n.parameters()]
        for step in range(3):
            for p in rp:
                p.grad = torch.zeros_like(p)
            tot = 0
            for r in range(world):
                for n in ref:
                    tot = tot + n(data[r, step]).pow(2).mean()
            (tot / world).backward()
            with torch.no_grad():
                for p in rp:
                    p -= 0.05 * p.grad
        out["ref"] = torch.cat([p.detach().reshape(-1) for p in rp])
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_bucket_layout_unused_heads_and_broadcast():
    """World size 8 over gloo: six per-network segments cut into several buckets each, unused ``fc`` heads inside the segments, a
    different initial state per rank (the broadcast must repair it), different data per rank, three optimiser steps.  Every replica
    must end bit-identical and on the single-process result over the 8 concatenated shards; every bucket that holds a used
    parameter must have left DURING the backward pass."""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), out), nprocs=world, join=True)
    p0, overl0, buckets = out[0]
    for r in range(1, world):
        assert torch.equal(out[r][0], p0), "replica %d diverged" % r
        assert out[r][1] == overl0 and out[r][2] == buckets
    assert torch.allclose(p0, out["ref"], rtol=2e-5, atol=1e-6), float((p0 - out["ref"]).abs().max())
    # layout: contiguous cover of the flat buffer, several buckets, none straddling a network; buckets made of unused heads only leave at finish()
    assert buckets[0][0] == 0 and all(buckets[i][1] == buckets[i + 1][0] for i in range(len(buckets) - 1))
    n_used_buckets = sum(1 for b in buckets if b[2] > 0)
    assert len(buckets) >= 8 and overl0 == [n_used_buckets] * 3


def test_trainer_bucket_layout_for_the_real_networks():
    """The bucket partition of the REAL parameter list (the six ResNet-18 networks of BASELINE config 2 / 4, built on the CPU: no
    kernels involved): ~25 MB buckets, none across a network boundary, the eight ``encoder.fc.*`` tensors not waited for, the whole
    197 MB gradient covered exactly once."""
    from fusiondepth_amd import dp, networks
    m = [networks.ResnetEncoder(18, False), networks.ResnetEncoder(18, False, beam_encoder=True),
         networks.ResnetEncoder(18, False, num_input_images=2, beam_encoder=True)]
    m.append(networks.DepthDecoder(m[0].num_ch_enc, range(4)))
    m.append(networks.ResnetEncoder(18, False, num_input_images=2))
    m.append(networks.PoseDecoder(m[4].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2))
    params = [p for n in m for p in n.parameters()]
    unused = [p for n in m for name, p in n.named_parameters() if name.startswith("encoder.fc.")]
    assert len(unused) == 8
    flat = dp.FlatParameters(params)
    sync = dp.GradientSynchronizer(flat, 8, segments=[len(list(n.parameters())) for n in m], never_used=unused)
    b = sync.buckets
    assert b[0][0] == 0 and b[-1][1] == flat.numel() and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    seg_ends, acc = [], 0
    for n in m:
        acc += sum(p.numel() for p in n.parameters())
        seg_ends.append(acc)
    ends = [x[1] for x in b]
    assert all(e in ends for e in seg_ends), "a bucket straddles two networks (their backward passes run on different streams)"
    assert sum(x[2] for x in b) == len(params) - 8
    sizes = [4 * (x[1] - x[0]) for x in b]
    assert max(sizes) < 2 * (25 << 20) + (10 << 20) and 6 <= len(b) <= 16, sizes
    assert flat.numel() == 49182752 + 4 * (512 * 1000 + 1000)          # SURVEY 8e's 49 182 752 trained parameters + the four unused fc heads


# ---- host placement ---------------------------------------------------------------------------------
def _fake_topology(root, n_gpus=8, nodes=2, cpus_per_node=96, smt=True):
    """An MI300X / MI355X-like host under ``root``: n_gpus AMD display devices, half per socket; each socket's cpulist = its physical
    cores followed by their hyper-thread siblings ("0-95,192-287"), as Linux prints it."""
    for g in range(n_gpus):
        d = os.path.join(root, "class/drm/card%d/device" % g)
        os.makedirs(d)
        open(os.path.join(d, "vendor"), "w").write("0x1002\n")
        open(os.path.join(d, "pci_address"), "w").write("0000:%02x:00.0\n" % (0x10 + 0x10 * g))
        open(os.path.join(d, "numa_node"), "w").write("%d\n" % (g * nodes // n_gpus))
    os.makedirs(os.path.join(root, "class/drm/card0-DP-1/device"))      # a connector entry: must be ignored
    total = nodes * cpus_per_node
    for n in range(nodes):
        d = os.path.join(root, "devices/system/node/node%d" % n)
        os.makedirs(d)
        lo = n * cpus_per_node
        txt = "%d-%d" % (lo, lo + cpus_per_node - 1)
        if smt:
            txt += ",%d-%d" % (total + lo, total + lo + cpus_per_node - 1)
        open(os.path.join(d, "cpulist"), "w").write(txt + "\n")
    for c in range(total * (2 if smt else 1)):
        d = os.path.join(root, "devices/system/cpu/cpu%d/topology" % c)
        os.makedirs(d)
        open(os.path.join(d, "thread_siblings_list"), "w").write(("%d,%d\n" % (c % total, c % total + total)) if smt else ("%d\n" % c))
    return list(range(total * (2 if smt else 1)))


def _worker_affinity(rank, world, port, root, online, out):
    from fusiondepth_amd import dp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = dp.affinity_plan(world, sysfs=root, online=online)[rank]             # every rank derives the same plan on its own
    sets = [None] * world
    dist.all_gather_object(sets, mine)
    if rank == 0:
        torch.save(sets, out)
    dist.destroy_process_group()


def test_eight_ranks_get_disjoint_cores_of_their_gpus_numa_node(tmp_path):
    """8 ranks on a faked 2-socket, 8-GPU topology: every rank's CPU set lies inside its GPU's NUMA node, the sets are pairwise disjoint
    and together use the node's cores evenly; a host with fewer cores than ranks is reported (overlapping sets)."""
    from fusiondepth_amd import dp
    root = str(tmp_path / "sys")
    online = _fake_topology(root)
    assert dp.gpu_numa_nodes(8, root) == [0, 0, 0, 0, 1, 1, 1, 1]
    out = str(tmp_path / "sets.pt")
    mp.spawn(_worker_affinity, args=(8, _free_port(), root, online, out), nprocs=8, join=True)
    sets = torch.load(out)
    node_cpus = [set(range(0, 96)) | set(range(192, 288)), set(range(96, 192)) | set(range(288, 384))]
    for r, cpus in enumerate(sets):
        assert len(cpus) == 48 and set(cpus) <= node_cpus[r // 4], (r, cpus[:4])
        for q in range(r):
            assert not (set(cpus) & set(sets[q])), "ranks %d and %d share cores" % (q, r)
    assert set().union(*[set(c) for c in sets[:4]]) == node_cpus[0]
    assert sets[0] == list(range(0, 24)) + list(range(192, 216)), "24 physical cores with their hyper-thread siblings"
    # a cgroup that leaves 2 cores for 4 ranks of a node: the plan still gives everybody a core, and says so through overlap
    tight = dp.affinity_plan(8, sysfs=root, online=[0, 1, 96, 97])            # (4 cores for 8 ranks)
    assert all(len(c) >= 1 for c in tight) and set(tight[0]) & set(tight[2])
    # unknown NUMA nodes (-1: VMs, single-socket hosts): the online cores are split among all local ranks
    root2 = str(tmp_path / "sys2")
    _fake_topology(root2, nodes=1, cpus_per_node=64, smt=False)
    for g in range(8):
        open(os.path.join(root2, "class/drm/card%d/device/numa_node" % g), "w").write("-1\n")
    flat = dp.affinity_plan(8, sysfs=root2, online=list(range(64)))
    assert sorted(c for s_ in flat for c in s_) == list(range(64)) and all(len(s_) == 8 for s_ in flat)
