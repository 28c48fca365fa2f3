"""Record a no-grad region's libfdhip calls once, replay them from ONE C call (``fd_replay``, csrc/replay.hip).

Why: the Python layer pays 10 - 18 us of interpreter / ctypes / autograd time per kernel launch.  The Refiner's frozen stage-1
networks (refiner.py:299-330: depth / LiDAR / pose encoders and the depth decoder under ``torch.no_grad()``) are ~400 such launches
per step that are the same calls on the same shapes every time - the step was host-bound.  A hipGraph of
the same region is no faster on ROCm 7.2 (its launch cost per kernel node equals the eager launch it replaces); this replays the
*calls*: a C loop over typed records, every launch on the caller's current stream like the eager path, so stream overlap, events and
everything outside the region stay as they are.

How: while ``Replayable`` runs its function eagerly in recording mode, ``_lib.call`` notes every entry-point call and ``_lib.ptr``
every tensor handed over.  Pointers are then classified - an INPUT of the region, a PERSISTENT tensor (parameters, buffers, cached
weight layouts, descriptors: recorded as literals, their owners kept alive and fingerprinted) or an INTERMEDIATE (everything else:
moved into one arena that is allocated once per replay).  The plan is validated before use: a replay on the recording's inputs must
reproduce the eager outputs BIT FOR BIT, otherwise (an ATen op inside the region, a host-side pointer table, a second stream) the
region stays eager and says why once.  A plan is dropped when a persistent tensor moves, a weight epoch changes, or ``fd_tuning``
changes.
"""
import ctypes
import struct
import warnings

import torch
from torch.utils import _pytree

from . import _lib
from . import tuning

MAX_ARGS = 24


class CallRec(ctypes.Structure):
    """Mirror of ``fd_call_rec`` (include/fdhip.h)."""
    _fields_ = [("fn", ctypes.c_int), ("nargs", ctypes.c_int), ("arg", ctypes.c_longlong * MAX_ARGS), ("kind", ctypes.c_ushort * MAX_ARGS)]


class NotRecordable(RuntimeError):
    pass


_FN_INDEX = {}


def _fn_table():
    if not _FN_INDEX:
        lib = _lib.load()
        for i in range(lib.fd_replay_function_count()):
            _FN_INDEX[lib.fd_replay_function_name(i).decode()] = (i, lib.fd_replay_function_signature(i).decode())
    return _FN_INDEX


class Recorder:
    """Active while a region runs eagerly; ``_lib.call`` / ``_lib.ptr`` report to it."""

    def __init__(self, mute_grad_ready=False):
        self.calls = []          # (name, args as passed)
        self.tensors = {}        # data_ptr -> tensor (kept alive until the plan is built)
        self.stream = None
        self.effects = []        # host-side effects the region has besides its calls: ("note_use" | "grad_ready" | "bn_counter" | "layout", payload)
        self.mute_grad_ready = mute_grad_ready

    def side(self, kind, payload):
        self.effects.append((kind, payload))

    def saw_tensor(self, t):
        self.tensors[t.data_ptr()] = t

    def saw_call(self, name, args):
        self.calls.append((name, args))


def _double_bits(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


class Plan:
    def __init__(self, recs, n_recs, arena_bytes, out_specs, out_tree, in_specs, keep, fingerprint, tally, stream_checked):
        self.recs, self.n_recs, self.arena_bytes = recs, n_recs, arena_bytes
        self.out_specs, self.out_tree, self.in_specs = out_specs, out_tree, in_specs
        self.keep, self.fingerprint, self.tally = keep, fingerprint, tally

    def replay(self, inputs, keep_arena=False):
        arena = torch.empty((max(self.arena_bytes, 256),), dtype=torch.uint8, device=inputs[0].device)
        ptrs = (ctypes.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
        _lib.call("fd_replay", ctypes.addressof(self.recs), self.n_recs, arena.data_ptr(), ctypes.addressof(ptrs), len(inputs), _lib.stream())
        outs = []
        for kind, off, shape, dtype, idx in self.out_specs:
            if kind == "none":
                outs.append(None)
            elif kind == "arena":
                n = 1
                for d in shape:
                    n *= d
                outs.append(arena[off:off + n * dtype.itemsize].view(dtype).view(shape))
            else:                                   # an output that IS an input (or a view of one)
                outs.append(inputs[idx])
        from . import functional as FD
        if FD.CONV_FLOP_TALLY is not None:
            FD.CONV_FLOP_TALLY[0] += self.tally
        res = _pytree.tree_unflatten(outs, self.out_tree) if self.out_tree is not None else outs
        return (res, arena) if keep_arena else res


def _storage_span(t):
    st = t.untyped_storage()
    return st.data_ptr(), st.nbytes()


def build_plan(rec, inputs, outputs_flat, out_tree, persistent, tally, foreign=None):
    """Classify the recorded pointers and lay the intermediates out in an arena -> Plan (raises NotRecordable).  ``foreign``: pointer ->
    (input slot, byte offset) or None, asked first (a backward plan finds the forward pass's saved tensors in the forward arena)."""
    table = _fn_table()
    in_spans = []
    for i, t in enumerate(inputs):
        base, nbytes = _storage_span(t)
        in_spans.append((base, base + nbytes, i, t.data_ptr()))
    pers_spans = []
    for t in persistent:
        base, nbytes = _storage_span(t)
        pers_spans.append((base, base + max(nbytes, 1)))
    pers_spans.sort()
    host_ok = _lib.HOST_PERSISTENT

    pers_hit = set()

    def in_persistent(p):
        for lo, hi in pers_spans:                  # (a few hundred spans, once per plan)
            if lo <= p < hi:
                pers_hit.add(lo)
                return True
        return False

    arena_of = {}                                   # storage base -> (arena offset, nbytes)
    arena_bytes = [0]

    def arena_offset(t, p):
        base, nbytes = _storage_span(t)
        if base not in arena_of:
            arena_of[base] = (arena_bytes[0], nbytes)
            arena_bytes[0] += (nbytes + 255) // 256 * 256
        return arena_of[base][0] + (p - base)

    stream0 = rec.stream
    recs = (CallRec * max(len(rec.calls), 1))()
    for k, (name, args) in enumerate(rec.calls):
        if name not in table:
            raise NotRecordable("%s is not a stream-ordered entry point" % name)
        fn, sig = table[name]
        if len(args) != len(sig) or len(sig) > MAX_ARGS:
            raise NotRecordable("%s: %d arguments recorded, signature has %d" % (name, len(args), len(sig)))
        r = recs[k]
        r.fn, r.nargs = fn, len(sig)
        for i, (kind, a) in enumerate(zip(sig, args)):
            if kind in "il":
                if not isinstance(a, int):
                    raise NotRecordable("%s argument %d: %r is not an integer" % (name, i, type(a)))
                r.arg[i], r.kind[i] = int(a), 0
            elif kind in "fd":
                r.arg[i], r.kind[i] = _double_bits(a), 0
            else:                                   # pointer
                if i == len(sig) - 1:               # the stream
                    if a != stream0:
                        raise NotRecordable("%s was issued on a second stream" % name)
                    r.arg[i], r.kind[i] = 0, 3
                    continue
                if a is None:
                    r.arg[i], r.kind[i] = 0, 0
                    continue
                if not isinstance(a, int):
                    raise NotRecordable("%s argument %d is a host object (%s): pointer tables are not recordable" % (name, i, type(a).__name__))
                t = rec.tensors.get(a)
                if t is None:
                    if a in host_ok:
                        r.arg[i], r.kind[i] = a, 0
                        continue
                    raise NotRecordable("%s argument %d: a pointer that did not come from a tensor" % (name, i))
                slot = foreign(a) if foreign is not None else None
                for lo, hi, idx, dptr in (in_spans if slot is None else ()):
                    if lo <= a < hi:
                        slot = (idx, a - dptr)
                        break
                if slot is not None:
                    if slot[1] < 0 or slot[0] > 4000:
                        raise NotRecordable("view of an input in front of its data pointer")
                    r.arg[i], r.kind[i] = slot[1], 2 | (slot[0] << 4)
                elif in_persistent(a):
                    r.arg[i], r.kind[i] = a, 0
                else:
                    r.arg[i], r.kind[i] = arena_offset(t, a), 1
    out_specs = []
    for t in outputs_flat:
        if t is None:
            out_specs.append(("none", 0, (), None, -1))
            continue
        p = t.data_ptr()
        base, _ = _storage_span(t)
        hit = None
        for lo, hi, idx, dptr in in_spans:
            if p == dptr and tuple(t.shape) == tuple(inputs[idx].shape):
                hit = idx
        if hit is not None:
            out_specs.append(("input", 0, tuple(t.shape), t.dtype, hit))
            continue
        if base not in arena_of or not t.is_contiguous():
            raise NotRecordable("an output of the region was not produced by a recorded call (or is not contiguous)")
        out_specs.append(("arena", arena_of[base][0] + (p - base), tuple(t.shape), t.dtype, -1))
    in_specs = [(tuple(t.shape), t.dtype) for t in inputs]
    used = [t for t in persistent if _storage_span(t)[0] in pers_hit]
    plan = Plan(recs, len(rec.calls), arena_bytes[0], out_specs, out_tree, in_specs, used, None, tally, True)
    plan.arena_map = arena_of
    return plan


class Replayable:
    """``Replayable(fn, persistent=lambda: tensors)(*input_tensors)``: ``fn`` runs eagerly the first ``warm`` times per input signature
    (weight layouts get derived), is recorded + validated on the next call, and replayed afterwards.  Must be called under
    ``torch.no_grad()`` with contiguous CUDA tensors; the outputs are views of a per-call arena."""

    def __init__(self, fn, persistent, name="region", warm=1):
        self.fn, self.persistent, self.name, self.warm = fn, persistent, name, warm
        self.plans, self.seen, self.disabled = {}, {}, None

    def _fingerprint(self, frozen_only=True):
        from . import functional as FD
        # trained weights inside the region: their layouts are refreshed once per optimiser step behind events this replay does not
        # see - such a plan is only good until the next optimiser step (i.e. useless, but never wrong)
        return (tuning.generation(), FD._FROZEN_EPOCH[0], 0 if frozen_only else FD._WEIGHTS_EPOCH[0])

    def __call__(self, *inputs):
        if self.disabled is not None or not tuning.host.replay_frozen or torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return self.fn(*inputs)
        key = tuple((tuple(t.shape), t.dtype, t.device.index, getattr(t, "_fd_normalized", False)) for t in inputs)
        ent = self.plans.get(key)
        if ent is not None:
            plan, fp, ptrs, frozen_only = ent
            if fp == self._fingerprint(frozen_only) and ptrs == [(t.data_ptr(), t._version) for t in plan.keep]:
                return plan.replay(inputs)
            del self.plans[key]                     # weights moved / changed / tuning changed: record again
            self.seen[key] = 0
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n < self.warm:
            return self.fn(*inputs)
        return self._record(key, inputs)

    def _record(self, key, inputs):
        from . import functional as FD
        for t in inputs:
            if not (t.is_cuda and t.is_contiguous()):
                return self.fn(*inputs)
        rec = Recorder()
        rec.stream = _lib.stream()
        tally0 = FD.CONV_FLOP_TALLY[0] if FD.CONV_FLOP_TALLY is not None else None
        had_tally = FD.CONV_FLOP_TALLY is not None
        if not had_tally:
            FD.CONV_FLOP_TALLY = [0.0]
            tally0 = 0.0
        _lib.RECORDER[0] = rec
        try:
            out = self.fn(*inputs)
        finally:
            _lib.RECORDER[0] = None
            tally = FD.CONV_FLOP_TALLY[0] - tally0
            if not had_tally:
                FD.CONV_FLOP_TALLY = None
        flat, tree = _pytree.tree_flatten(out)
        try:
            if not all(t is None or torch.is_tensor(t) for t in flat):
                raise NotRecordable("the region returns non-tensor leaves")
            persistent = [t for t in self.persistent() if torch.is_tensor(t) and t.is_cuda]
            persistent += [e[1] for e in FD._WT_CACHE.values()] + FD.folded_tensors()
            plan = build_plan(rec, list(inputs), flat, tree, persistent, tally)
            again = plan.replay(list(inputs))
            flat2, _ = _pytree.tree_flatten(again)
            torch.cuda.current_stream().synchronize()
            for a, b in zip(flat, flat2):
                if a is None:
                    continue
                if a.shape != b.shape or not torch.equal(a, b):
                    if not (torch.isnan(a) == torch.isnan(b)).all() or not torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)):
                        raise NotRecordable("the replay does not reproduce the eager outputs bit for bit (an operation outside libfdhip "
                                            "inside the region?)")
        except NotRecordable as e:
            self.disabled = str(e)
            warnings.warn("fusiondepth_amd.replay: %s stays on the eager path: %s" % (self.name, e))
            return out
        frozen_only = all(getattr(t, "_fd_frozen", False) or not isinstance(t, torch.nn.Parameter) or t.dim() != 4 for t in self.persistent())
        self.plans[key] = (plan, self._fingerprint(frozen_only), [(t.data_ptr(), t._version) for t in plan.keep], frozen_only)
        return out


# ------------------------------------------------------------------------------------------------------------------ training
class _ReplayedNet(torch.autograd.Function):
    """ONE autograd node for a whole network: forward = its recorded calls replayed into a fresh arena (which is what the backward's
    saved tensors live in), backward = the recorded calls of its backward pass.  Parameter gradients accumulate straight into the flat
    gradient buffer (functional.enable_direct_grad), so the node returns no gradient at all for them."""

    @staticmethod
    def forward(ctx, state, x, anchor):
        from . import functional as FD
        if state.late_f:
            FD._wait_late_layouts()
        for kind, payload in state.fwd_effects:
            if kind == "note_use":
                FD._note_use(*payload)
            elif kind == "bn_counter":
                FD.bump_bn_counter(*payload)
        outs, arena = state.fwd.replay([x], keep_arena=True)
        ctx.state, ctx.arena, ctx.x = state, arena, x
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        from . import functional as FD
        st = ctx.state
        got = tuple(i for i, g in enumerate(gouts) if g is not None)
        if not set(got) <= set(st.pattern):
            # a gradient at an output the recorded backward pass does not start from would be dropped silently
            raise RuntimeError("fusiondepth_amd.replay: %s was recorded with gradients arriving at outputs %s, this backward pass brings %s "
                               "(set tuning.host.replay_train = False for this graph)" % (st.name, st.pattern, got))
        # an output of the pattern without a gradient this time (a loss that skips a scale): zeros give the same sums
        gs = [_lib.f32(gouts[i]) if gouts[i] is not None else torch.zeros(st.out_shapes[i], device=ctx.x.device) for i in st.pattern]
        if st.late_b or st.late_f:
            FD._wait_late_layouts()
        out = st.bwd.replay(gs + [ctx.x, ctx.arena])
        ctx.arena = None
        for kind, payload in st.bwd_effects:
            if kind == "grad_ready":
                FD._grad_ready(*payload)
        return None, (out[0] if out else None), None


class _TrainState:
    def __init__(self, name):
        self.name, self.calls, self.fwd, self.bwd = name, 0, None, None
        self.seen_grads, self.pattern = set(), None
        self.fwd_effects, self.bwd_effects, self.layouts = [], [], []
        self.late_f = self.late_b = True


class TrainReplayable:
    """A network's TRAINING forward + backward as two recorded call sequences behind one autograd node.

    ``TrainReplayable(net, call, name)(x)``: the first calls per input signature run eagerly (and note which outputs receive a
    gradient); then one call is recorded - it is still the real, eager forward of that step - followed by an isolated recording of the
    backward pass on random gradients (its parameter gradients are thrown away again); both sequences are validated bit for bit
    (outputs, BatchNorm buffers, parameter gradients) before the node replaces the eager graph.  Requirements: parameters with direct
    gradient accumulation, one stream, no operation outside libfdhip inside the network - otherwise it stays eager and says why."""
    WARM = 2

    def __init__(self, modules, call, name, key_extra=lambda: None):
        self.modules, self.call, self.name, self.key_extra = list(modules), call, name, key_extra
        self.states, self.disabled = {}, None
        self.params = [p for m in self.modules for p in m.parameters()]
        self.anchor = next((p for p in self.params if p.requires_grad), None)

    @property
    def training(self):
        return all(m.training for m in self.modules)

    def _eager(self, st, x):
        outs = self.call(x)
        idx = st.seen_grads
        for i, t in enumerate(outs):
            if t is not None and t.requires_grad:
                t.register_hook(lambda g, i=i: idx.add(i))
        return outs

    def __call__(self, x):
        from . import functional as FD
        if (self.disabled is not None or not tuning.host.replay_train or not torch.is_grad_enabled() or not self.training
                or self.anchor is None or torch.cuda.is_current_stream_capturing() or not (x.is_cuda and x.is_contiguous())):
            return self.call(x)
        key = (tuple(x.shape), x.dtype, getattr(x, "_fd_normalized", False), bool(x.requires_grad), FD._BN_GROUPS[0], tuning.generation(),
               self.key_extra())
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = _TrainState(self.name)
        if st.fwd is None:
            st.calls += 1
            if st.calls <= self.WARM:
                return self._eager(st, x)
            return self._record(st, x)
        late_f = late_b = False
        for w, k, ptr_ in st.layouts:                         # every weight layout the recorded calls read must be current
            ent = FD._WT_CACHE.get(k)
            if ent is None or ent[0] != FD._layout_stamp(w) or ent[1].data_ptr() != ptr_:
                return self.call(x)                           # (the eager call re-derives it; the plan stays valid for the next step)
            ent[5] = FD._WEIGHTS_EPOCH[0]
            if ent[4]:                                        # refreshed on the side stream behind Adam: its readers wait for that launch
                if k[1] == "f":
                    late_f = True
                else:
                    late_b = True
        # (only the segments that READ such a layout wait: the stems and first blocks start beside the re-layout launch, like the eager path)
        st.late_f, st.late_b = late_f, late_b
        return list(_ReplayedNet.apply(st, x, self.anchor))

    # -- recording -------------------------------------------------------------------------------------------------------------
    def _buffers(self):
        return [b for m in self.modules for b in m.buffers() if b.is_floating_point()]

    def _record(self, st, x):
        from . import functional as FD
        try:
            return self._record_inner(st, x, FD)
        except NotRecordable as e:
            self.disabled = str(e)
            warnings.warn("fusiondepth_amd.replay: %s stays on the eager path: %s" % (self.name, e))
            return self.call(x) if not hasattr(e, "outs") else e.outs

    def _record_inner(self, st, x, FD):
        if not all(getattr(p, "_fd_direct_grad", False) and p.grad is not None for p in self.params if p.requires_grad):
            raise NotRecordable("parameters without direct gradient accumulation")
        bufs = self._buffers()
        before = [b.clone() for b in bufs]
        rec = Recorder()
        rec.stream = _lib.stream()
        _lib.RECORDER[0] = rec
        try:
            outs = self.call(x)                               # the real forward of this step (eager graph)
        finally:
            _lib.RECORDER[0] = None
        def fail(msg):
            e = NotRecordable(msg)
            e.outs = outs
            return e
        after = [b.clone() for b in bufs]
        pattern = tuple(sorted(st.seen_grads))
        flat = list(outs)
        persistent = [t for t in self.params + [b for m in self.modules for b in m.buffers()] if t.is_cuda]
        persistent += [p.grad for p in self.params if p.grad is not None]
        persistent += [e[1] for e in FD._WT_CACHE.values()]
        try:
            fwd = build_plan(rec, [x], flat, None, persistent, 0.0)
        except NotRecordable as e:
            raise fail(str(e))
        # validate the forward: BatchNorm buffers back to their state before this step's forward, replay, compare outputs + buffers
        with torch.no_grad():
            for b, v in zip(bufs, before):
                b.copy_(v)
            again, arena = fwd.replay([x], keep_arena=True)
            ok = all((a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and torch.equal(a, b)) for a, b in zip(flat, again))
            ok = ok and all(torch.equal(b, v) for b, v in zip(bufs, after))
            for b, v in zip(bufs, after):
                b.copy_(v)
        if not ok:
            raise fail("the forward replay does not reproduce the eager outputs / BatchNorm buffers bit for bit")
        # the backward pass, recorded in isolation on random gradients (the gradient pattern of the warm-up steps)
        if not pattern or any(flat[i] is None or not flat[i].requires_grad for i in pattern):
            raise fail("no output of the network received a gradient in the warm-up steps")
        touched = [p.grad for p in self.params if p.grad is not None]
        saved = [g.clone() for g in touched]                  # (zero at this point of a step, but a caller may accumulate)
        gen = torch.Generator(device=x.device).manual_seed(1)
        gs = [torch.randn(flat[i].shape, device=x.device, generator=gen) for i in pattern]
        rec2 = Recorder(mute_grad_ready=True)
        rec2.stream = _lib.stream()
        _lib.RECORDER[0] = rec2
        try:
            # (every trainable parameter is asked for, so that each node's needs_input_grad is what loss.backward() makes it)
            wrt = ([x] if x.requires_grad else []) + [p for p in self.params if p.requires_grad]
            got = torch.autograd.grad([flat[i] for i in pattern], wrt, gs, retain_graph=True, allow_unused=True)
            torch.cuda.current_stream().synchronize()         # (the engine's worker thread has issued everything)
        finally:
            _lib.RECORDER[0] = None
        gx = got[0] if x.requires_grad else None
        if x.requires_grad and gx is None:
            raise fail("the input requires a gradient but the backward pass produced none")
        want = [g.clone() for g in touched]
        amap = fwd.arena_map
        fwd_tensors = rec.tensors

        n_g = len(gs)

        def foreign_slot(p):                                  # a tensor the forward pass saved: it lives in the forward arena
            t = fwd_tensors.get(p)
            if t is None:
                return None
            base, _ = _storage_span(t)
            hit = amap.get(base)
            return (n_g + 1, hit[0] + (p - base)) if hit is not None else None      # inputs of the backward plan: gradients..., x, arena

        try:
            bwd = build_plan(rec2, gs + [x], [gx] if gx is not None else [], None, persistent, 0.0, foreign=foreign_slot)
        except NotRecordable as e:
            with torch.no_grad():
                for g, v in zip(touched, saved):
                    g.copy_(v)
            raise fail("backward: " + str(e))
        with torch.no_grad():                                 # validate: same gradients from the replayed backward on the replayed forward
            for g, v in zip(touched, saved):
                g.copy_(v)
            gx2 = bwd.replay(gs + [x, arena])
            ok = all(torch.equal(g, w_) for g, w_ in zip(touched, want)) and (gx is None or torch.equal(gx2[0], gx))
            for g, v in zip(touched, saved):
                g.copy_(v)
        del arena
        if not ok:
            raise fail("the backward replay does not reproduce the eager parameter gradients bit for bit")
        st.fwd, st.bwd, st.pattern = fwd, bwd, pattern
        st.out_shapes = [tuple(t.shape) if t is not None else None for t in flat]
        st.fwd_effects = [e for e in rec.effects if e[0] in ("note_use", "bn_counter")]
        st.bwd_effects = [e for e in rec2.effects if e[0] == "grad_ready"]
        lay = {}
        for kind, payload in rec.effects + rec2.effects:
            if kind == "layout":
                w, k = payload
                ent = FD._WT_CACHE.get(k)
                if ent is not None:
                    lay[k] = (w, k, ent[1].data_ptr())
        st.layouts = list(lay.values())
        return outs


class ReplayedEncoder:
    """A ResnetEncoder's training pass as FIVE replayed segments - stem, layer1 .. layer4 - each one autograd node: the gradients that
    arrive at the intermediate feature maps (the decoder's skip connections) are summed with the chain's own by autograd BETWEEN the
    segments, exactly as in the eager graph, so no operation outside libfdhip falls inside a recorded sequence."""

    def __init__(self, net, name):
        e = net.encoder
        self.net = net
        self.stem = TrainReplayable([_Holder(e.conv1, e.bn1)], lambda x: list(net.stem(x)), name + ".stem",
                                    key_extra=lambda: (net.stem_feature_needed, tuning.host.fused_stem_tail))
        self.layers = [TrainReplayable([getattr(e, "layer%d" % i)], (lambda x, m=getattr(e, "layer%d" % i): [m(x)]), "%s.layer%d" % (name, i))
                       for i in range(1, 5)]

    def segments(self):
        return [self.stem] + self.layers

    def __call__(self, x):
        f0, x = self.stem(x)
        feats = [f0]
        for seg in self.layers:
            x = seg(x)[0]
            feats.append(x)
        self.net.features = feats
        return feats


class _Holder:
    """the stem's two modules as one parameter / buffer / training-flag scope"""

    def __init__(self, *mods):
        self._mods = mods

    def parameters(self):
        for m in self._mods:
            yield from m.parameters()

    def buffers(self):
        for m in self._mods:
            yield from m.buffers()

    @property
    def training(self):
        return all(m.training for m in self._mods)
