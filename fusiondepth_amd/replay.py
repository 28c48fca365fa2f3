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

    def __init__(self):
        self.calls = []          # (name, args as passed)
        self.tensors = {}        # data_ptr -> tensor (kept alive until the plan is built)
        self.stream = None

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

    def replay(self, inputs):
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
        return _pytree.tree_unflatten(outs, self.out_tree)


def _storage_span(t):
    st = t.untyped_storage()
    return st.data_ptr(), st.nbytes()


def build_plan(rec, inputs, outputs_flat, out_tree, persistent, tally):
    """Classify the recorded pointers and lay the intermediates out in an arena -> Plan (raises NotRecordable)."""
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
                slot = None
                for lo, hi, idx, dptr in in_spans:
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
    return Plan(recs, len(rec.calls), arena_bytes[0], out_specs, out_tree, in_specs, used, None, tally, True)


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
            persistent += [e[1] for e in FD._WT_CACHE.values()]
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
