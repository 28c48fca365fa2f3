"""Data-parallel plumbing (new w.r.t. the reference, which is single-GPU — SURVEY.md §2.2/§8e).

One process per GPU.  All trainable tensors live in ONE flat fp32 buffer (and their gradients in another), so
  * the optimiser is a single fused Adam launch over the flat buffer, and
  * the gradient exchange is a few large RCCL all-reduces over contiguous slices ("buckets", ~25 MB) issued
    as soon as autograd has produced every gradient of a bucket (post-accumulate hooks), overlapping the rest
    of the backward pass.  xGMI is point-to-point, so few large messages beat many small ones.
BatchNorm statistics stay per replica (the reference has no SyncBN).  Gradients are summed; the 1/world
averaging is folded into the Adam kernel's ``grad_scale``.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/WORLD_SIZE/MASTER_*) if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("FD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class FlatParameters:
    """Re-homes ``params`` into one contiguous buffer; ``.grad`` of each becomes a view of ``flat_grad``."""

    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in sizes:
            self.offsets.append(self.offsets[-1] + n)
        total = self.offsets[-1]
        self.flat_param = torch.empty(total, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(total, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, sizes):
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):     # autograd may have replaced .grad; re-attach the views
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def numel(self):
        return self.flat_param.numel()


class GradientSynchronizer:
    """Bucketed, backward-overlapped all-reduce(sum) of a FlatParameters' gradient buffer."""

    def __init__(self, flat, world_size, bucket_bytes=25 << 20, group=None):
        self.flat, self.world, self.group = flat, world_size, group
        self.armed = False
        self.handles = []
        self.buckets = []           # [start, end, n_params]
        self.param_bucket = []
        per = max(bucket_bytes // 4, 1)
        start, count = 0, 0
        for i, p in enumerate(flat.params):
            self.param_bucket.append(len(self.buckets))
            count += 1
            end = flat.offsets[i + 1]
            if end - start >= per or i == len(flat.params) - 1:
                self.buckets.append([start, end, count])
                start, count = end, 0
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        if world_size > 1:
            for i, p in enumerate(flat.params):
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            if not self.armed:
                return
            o, n = self.flat.offsets[i], param.numel()
            if param.grad is not None and param.grad.data_ptr() != self.flat.flat_grad.data_ptr() + 4 * o:
                # autograd assigned a fresh tensor instead of accumulating into the view: fold it back
                self.flat.flat_grad[o:o + n].copy_(param.grad.reshape(-1))
                param.grad = self.flat.flat_grad[o:o + n].view(param.shape)
            b = self.param_bucket[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e, _ = self.buckets[b]
        self.handles.append(dist.all_reduce(self.flat.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def arm(self):
        """Call right before the backward pass whose gradients complete the accumulation window."""
        self.armed = self.world > 1
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []

    def finish(self):
        """After backward: reduce buckets whose hooks never fired (unused parameters, e.g. the ResNet ``fc``)
        and wait for everything.  Returns the factor that turns the summed gradient into the mean."""
        if self.world > 1 and self.armed:
            if not any(self.launched):
                # no hook fired (the trainer's kernels accumulate gradients in place, autograd never sees them): the whole
                # flat buffer goes out as ONE all-reduce - the largest message the ring over xGMI can get
                dist.all_reduce(self.flat.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            else:
                for b in range(len(self.buckets)):
                    self._launch(b)
                for h in self.handles:
                    h.wait()
        self.armed = False
        return 1.0 / self.world


def broadcast_module_state(modules, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
