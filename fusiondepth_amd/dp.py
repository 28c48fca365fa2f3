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


# ---- host placement -----------------------------------------------------------------------------------------------------------
# One process per GPU issues ~830 launches per step from Python; the step is within 1.2 - 1.6x of that issue time, so a rank whose
# threads wander over both sockets (or share cores with another rank) becomes the straggler every all-reduce waits for.  Each rank is
# pinned to cores of ITS GPU's NUMA node, the node's cores divided among the ranks whose GPUs hang off it.
def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _parse_cpulist(text):
    cpus = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(n_gpus, sysfs="/sys", pci_ids=None):
    """NUMA node of local GPU 0 .. n_gpus-1 (-1: unknown).  ``pci_ids`` (``"0000:c1:00.0"`` per device, from the HIP runtime) name the
    devices exactly; without them the AMD display-class functions under ``<sysfs>/class/drm/card*/device`` are taken in PCI-address
    order - the order HIP enumerates them in when HIP_VISIBLE_DEVICES does not permute it."""
    if pci_ids:
        return [int(_read(os.path.join(sysfs, "bus/pci/devices", a, "numa_node")) or -1) for a in pci_ids[:n_gpus]]
    import glob
    cards = {}
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card*/device")):
        if "-" in os.path.basename(os.path.dirname(dev)):            # connectors (card0-DP-1)
            continue
        vendor = _read(os.path.join(dev, "vendor"))
        if vendor is not None and vendor.lower() != "0x1002":
            continue
        addr = os.path.basename(os.path.realpath(dev)) if os.path.islink(dev) else (_read(os.path.join(dev, "pci_address")) or os.path.dirname(dev))
        cards[addr] = int(_read(os.path.join(dev, "numa_node")) or -1)
    nodes = [cards[a] for a in sorted(cards)]
    return (nodes + [-1] * n_gpus)[:n_gpus]


def affinity_plan(local_world, sysfs="/sys", pci_ids=None, online=None):
    """-> one sorted CPU list per local rank: the cores of the rank's GPU's NUMA node, split evenly among the ranks on that node
    in whole physical cores (hyper-thread siblings stay with their core); ranks with an unknown node share ``online`` (default:
    this process's current affinity) the same way.  Pure function of the sysfs tree: testable on a faked topology."""
    nodes = gpu_numa_nodes(local_world, sysfs, pci_ids)
    if online is None:
        online = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    allowed = set(online)
    plan = [None] * local_world
    for node in sorted(set(nodes)):
        ranks = [r for r in range(local_world) if nodes[r] == node]
        cpus = _parse_cpulist(_read(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node))) if node >= 0 else []
        cpus = [c for c in cpus if c in allowed] or sorted(allowed)
        # physical cores = groups of hyper-thread siblings (Linux lists a socket as "0-95,192-287": cores, then their siblings):
        # a rank gets whole cores, never the sibling threads of another rank's cores
        groups, seen = [], {}
        for c in cpus:
            sib = _parse_cpulist(_read(os.path.join(sysfs, "devices/system/cpu/cpu%d/topology/thread_siblings_list" % c))) or [c]
            key = min(sib)
            if key not in seen:
                seen[key] = len(groups)
                groups.append([])
            groups[seen[key]].append(c)
        per = max(len(groups) // len(ranks), 1)
        for i, r in enumerate(ranks):
            lo = (i * per) % len(groups)
            plan[r] = sorted(c for grp in groups[lo:lo + per] for c in grp)
    return plan


def pin_to_gpu_numa(local_rank, local_world, sysfs="/sys", verbose=False):
    """Pin this process (all its threads: called before the autograd / RCCL threads exist) to its share of its GPU's NUMA node.
    Returns the CPU set, or None when the platform cannot (no sched_setaffinity) or FD_NO_AFFINITY is set."""
    if os.environ.get("FD_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    pci = None
    try:
        if torch.cuda.is_available():
            pci = []
            for i in range(min(local_world, torch.cuda.device_count())):
                p = torch.cuda.get_device_properties(i)
                pci.append("%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id))
            if len(pci) < local_world:
                pci = None
    except Exception:
        pci = None
    plan = affinity_plan(local_world, sysfs, pci)
    mine = plan[local_rank]
    shared = [r for r in range(local_world) if r != local_rank and set(plan[r]) & set(mine)]
    if shared:
        import warnings
        warnings.warn("fusiondepth_amd.dp: rank %d shares host cores with local ranks %s (%d cores for %d ranks on its NUMA node) - the "
                      "step is host-sensitive, expect stragglers" % (local_rank, shared, len(mine), local_world))
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    if verbose:
        print("fusiondepth_amd.dp: local rank %d pinned to %d cores (%d..%d)" % (local_rank, len(mine), mine[0], mine[-1]))
    return mine


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/WORLD_SIZE/MASTER_*) if world > 1; with more than one rank on this
    host each rank is pinned to cores of its GPU's NUMA node first (``pin_to_gpu_numa``)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1 and local_world > 1 and not dist.is_initialized():
        pin_to_gpu_numa(local_rank, local_world)
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
        from . import functional as FD
        FD.join_wgrad_streams()          # side-stream weight gradients still accumulating into the buffer come first
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):     # autograd may have replaced .grad; re-attach the views
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def numel(self):
        return self.flat_param.numel()


class GradientSynchronizer:
    """Bucketed, backward-overlapped all-reduce(sum) of a FlatParameters' gradient buffer.

    Buckets are contiguous slices of the flat buffer (>= ``bucket_bytes``, never across a ``segments`` boundary: the trainer
    passes one segment per network because each network's backward runs on its own HIP stream).  A bucket goes out as soon as
    the kernel that completes its last gradient has been LAUNCHED: the asynchronous collective is ordered behind that kernel
    through the stream the backward node runs on (RCCL waits on an event of the current stream), so it overlaps with the rest
    of the backward pass that is still being issued.  Two notification paths: autograd's post-accumulate hooks (gradients that
    autograd produces) and ``functional.add_grad_ready_callback`` (gradients the HIP kernels accumulate in place).  Few large
    messages by design: xGMI is point-to-point, a ring all-reduce is bound by one link (~153 GB/s), not by a switch."""

    def __init__(self, flat, world_size, bucket_bytes=25 << 20, group=None, segments=None, overlap=None, never_used=()):
        self.flat, self.world, self.group = flat, world_size, group
        self.armed = False
        self.handles = []
        self.buckets = []           # [start, end, n_params]
        self.param_bucket = []
        if overlap is None:
            from . import tuning
            overlap = tuning.host.dp_overlap
        self.overlap = bool(overlap)
        self.skip_comm = False      # bench only: measure a step without the exchange
        per = max(bucket_bytes // 4, 1)
        ends = set()                # parameter indices after which a bucket must close
        if segments:
            acc = 0
            for n in segments:
                acc += n
                ends.add(acc - 1)
        skip = set(id(p) for p in never_used)   # parameters that never receive a gradient (the ResNet `fc` heads): not waited for
        start, count = 0, 0
        for i, p in enumerate(flat.params):
            self.param_bucket.append(len(self.buckets))
            count += 0 if id(p) in skip else 1
            end = flat.offsets[i + 1]
            if end - start >= per or i == len(flat.params) - 1 or i in ends:
                self.buckets.append([start, end, count])
                start, count = end, 0
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.index = {id(p): i for i, p in enumerate(flat.params)}
        self.n_overlapped = 0       # buckets of the last window that went out before finish()
        self.done = {}              # parameter index -> in-place gradient kernels launched in this window
        self.late = []
        self.seen = set()
        if world_size > 1:
            for i, p in enumerate(flat.params):
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._make_hook(i))
            from . import functional as FD
            FD.add_grad_ready_callback(self._on_direct_grad)

    def _arrived(self, i):
        # Two notification paths can report the same parameter: the in-place kernels' callback, and autograd's post-accumulate
        # hook - which this torch also runs for a leaf whose Function returned no gradient (it fires once all uses of the leaf
        # have run their backward, i.e. never before the callback's last kernel).  Count each parameter once per window.
        if i in self.seen:
            return
        self.seen.add(i)
        b = self.param_bucket[i]
        self.pending[b] -= 1
        if self.pending[b] == 0 and self.overlap and not self.skip_comm:
            self._launch(b)
            self.n_overlapped += 1

    def _on_direct_grad(self, param):
        """functional._grad_ready: a kernel that accumulates into ``param.grad`` (a view of the flat buffer) has been launched;
        the gradient is complete once that has happened as often as the parameter was used in the forward pass."""
        if self.armed:
            i = self.index.get(id(param))
            if i is not None:
                from . import functional as FD
                if self.launched[self.param_bucket[i]]:
                    self.late.append((i, self.done.get(i, 0), FD.param_uses(param)))   # a kernel AFTER its bucket went out
                self.done[i] = self.done.get(i, 0) + 1
                if self.done[i] == FD.param_uses(param):
                    self._arrived(i)

    def _make_hook(self, i):
        def hook(param):
            if not self.armed:
                return
            o, n = self.flat.offsets[i], param.numel()
            if param.grad is not None and param.grad.data_ptr() != self.flat.flat_grad.data_ptr() + 4 * o:
                # autograd assigned a fresh tensor instead of accumulating into the view: fold it back
                self.flat.flat_grad[o:o + n].copy_(param.grad.reshape(-1))
                param.grad = self.flat.flat_grad[o:o + n].view(param.shape)
            self._arrived(i)
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
        self.n_overlapped = 0
        self.done = {}
        self.late = []
        self.seen = set()

    def finish(self):
        """After backward (and after the module streams were joined): reduce what has not gone out yet - buckets with
        parameters that received no gradient (the ResNet ``fc`` heads), or everything as ONE message when overlap is off -
        and make the current stream wait for every collective.  Returns the factor that turns the sum into the mean."""
        if self.world > 1 and self.armed and not self.skip_comm:
            if not any(self.launched):
                dist.all_reduce(self.flat.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            else:
                for b in range(len(self.buckets)):
                    self._launch(b)
                for h in self.handles:
                    h.wait()
        self.armed = False
        if self.late:
            raise RuntimeError("GradientSynchronizer: %d gradient kernels were launched after their bucket's all-reduce (first: "
                               "(parameter, kernels so far, announced uses) %s) - a parameter is used more often in the backward pass "
                               "than the forward pass announced" % (len(self.late), self.late[:6]))
        return 1.0 / self.world


def broadcast_module_state(modules, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
    # the writes went through .data: torch's version counters did not move, so the cached kernel-side weight layouts must be told
    from . import functional as FD
    FD.bump_weights_epoch()
    FD.invalidate_frozen_layouts()
