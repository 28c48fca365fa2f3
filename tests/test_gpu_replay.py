"""fd_replay (csrc/replay.hip) + fusiondepth_amd/replay.py: a no-grad region's libfdhip calls recorded once and replayed from ONE C call
- the Refiner's frozen stage-1 networks (refiner.py:299-330).  The replay must be the eager path bit for bit."""
import warnings

import numpy as np
import pytest
import torch

import inputs as gin

pytestmark = pytest.mark.gpu


def _encoder(layers=18, **kw):
    from fusiondepth_amd import functional as FD, networks
    torch.manual_seed(3)
    net = networks.ResnetEncoder(layers, False, **kw).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.1, 0.1)
    for p in net.parameters():
        p.requires_grad_(False)
    FD.enable_weight_cache(list(net.parameters()), frozen=True)
    return net


@pytest.mark.parametrize("layers", [18, 50])
def test_replayed_encoder_is_the_eager_encoder_bit_for_bit(layers):
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.replay import Replayable
    net = _encoder(layers)
    rp = Replayable(lambda x: list(net(x)), lambda: list(net.parameters()) + list(net.buffers()), name="test encoder")
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for k in range(5):                                    # call 0 eager (layouts), call 1 recorded + validated, 2.. replayed
            x = torch.rand(2, 3, 64, 96, device="cuda", generator=g)
            got = rp(x)
            want = list(net(x))
            assert len(got) == len(want) == 5
            for a, b in zip(got, want):
                assert torch.equal(a, b), "call %d" % k
        assert rp.disabled is None and len(rp.plans) == 1
        plan = next(iter(rp.plans.values()))[0]
        assert plan.n_recs >= 30 and plan.arena_bytes > 0          # (~45 calls before the frozen BatchNorms were folded into their convolutions)
        x2 = torch.rand(1, 3, 96, 160, device="cuda", generator=g)         # another shape: its own plan
        for k in range(3):
            for a, b in zip(rp(x2), net(x2)):
                assert torch.equal(a, b)
        assert len(rp.plans) == 2
        # the weights change behind torch's back (p.data.copy_ + FD.invalidate_frozen_layouts): the plan is dropped, results follow
        for p in net.parameters():
            p.data.mul_(1.25)
        FD.invalidate_frozen_layouts()
        x3 = torch.rand(2, 3, 64, 96, device="cuda", generator=g)
        for k in range(3):
            for a, b in zip(rp(x3), net(x3)):
                assert torch.equal(a, b), "after the weight change, call %d" % k


def test_region_with_an_operation_outside_libfdhip_stays_eager():
    from fusiondepth_amd.replay import Replayable
    net = _encoder(18)
    fn = lambda x: [f * 2.0 for f in net(x)]                  # an ATen multiply inside the region: not recordable
    rp = Replayable(fn, lambda: list(net.parameters()) + list(net.buffers()), name="test region")
    with torch.no_grad():
        x = torch.rand(2, 3, 64, 96, device="cuda")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for k in range(4):
                for a, b in zip(rp(x), fn(x)):
                    assert torch.equal(a, b)
        assert rp.disabled is not None and any("stays on the eager path" in str(m.message) for m in w)


def test_refiner_steps_with_replayed_frozen_networks_are_bit_identical():
    """two optimiser steps of the Refiner with the frozen networks replayed vs issued eagerly: same losses, same refine-decoder
    parameters, bit for bit; the replay really is in use (plans exist for encoder / beam_encoder / depth / the pose encoders)"""
    from fusiondepth_amd import tuning
    from test_gpu_refiner import _make
    B, H, W = 1, 192, 640
    batches = []
    for step in range(3):
        inp, noise = gin.refiner_inputs(880 + step, B, H, W)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        batches.append(g)
    res = {}
    for mode in (False, True):
        tuning.host.replay_frozen = mode
        try:
            rf, _, _ = _make(B, H, W)
            losses = []
            for b in batches:
                lg = rf.train_step({k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in b.items()})
                losses.append(float(lg["loss"]))
            torch.cuda.synchronize()
            res[mode] = (losses, torch.cat([p.detach().reshape(-1) for p in rf.models["refine2d_decoder"].parameters()]).clone(),
                         {k: (r.disabled, len(r.plans)) for k, r in rf._replays.items()})
        finally:
            tuning.host.replay_frozen = True
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1])
    used = res[True][2]
    for name in ("encoder", "beam_encoder", "depth", "pose_encoder", "beam_encoder_pose"):
        assert name in used and used[name] == (None, 1), (name, used)


@pytest.mark.parametrize("layers,H,W,bs", [(18, 128, 192, 4), (50, 64, 96, 2)])
def test_training_steps_with_replayed_encoders_are_bit_identical(layers, H, W, bs):
    """six optimiser steps of the Trainer with the four encoders' forward + backward behind replayed call sequences (from the third
    step on) against the same steps issued from Python: every loss and every parameter bit for bit, BatchNorm buffers included;
    the replay really is in use ('bit-identical parameters vs the Python-issued step')"""
    from fusiondepth_amd import tuning
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    from test_gpu_trainer import _batch
    res = {}
    for mode in (False, True):
        tuning.host.replay_train = mode
        try:
            torch.manual_seed(5)
            opt = MonodepthOptions().parse(["--num_layers", str(layers), "--weights_init", "scratch", "--batch_size", str(bs), "--height", str(H),
                                            "--width", str(W)])
            tr = Trainer(opt, verbose=False)
            for k, m in tr.models.items():
                gin.fill_params(m, 100 + len(k))
            from fusiondepth_amd import functional as FD
            FD.bump_weights_epoch()
            losses = []
            for step in range(6):
                mbs = []
                for g in range(tr.accumulate_step):
                    inp, noise = _batch(tr.batch_size, H, W, 300 + 10 * step + g)
                    b = {k: v.cuda() for k, v in inp.items()}
                    b["_noise"] = [n.cuda() for n in noise]
                    mbs.append(b)
                lg = tr.train_step(mbs if tr.accumulate_step > 1 or not tr.stack_microbatches else mbs)
                losses.append(float(lg["loss"].detach()))
            torch.cuda.synchronize()
            bufs = torch.cat([b.detach().reshape(-1).float() for m in tr.models.values() for b in m.buffers()])
            reps = getattr(tr, "_train_replays", {})
            res[mode] = (losses, tr.flat.flat_param.clone(), bufs.clone(),
                         {seg.name: (seg.disabled, [(s.fwd is not None, s.pattern) for s in seg.states.values()]) for r in reps.values() for seg in r.segments()})
            del tr
        finally:
            tuning.host.replay_train = True
    assert res[True][3], "the encoders were not routed through TrainReplayable"
    for name, (disabled, states) in res[True][3].items():
        assert disabled is None, (name, disabled)
        assert states and all(ok for ok, _ in states), (name, states)
    assert np.array_equal(np.array(res[False][0]), np.array(res[True][0]), equal_nan=True), (res[False][0], res[True][0])     # (NaN: an empty SI-log mask)
    assert torch.equal(res[False][1], res[True][1]), "parameters differ"
    assert torch.equal(res[False][2], res[True][2]), "BatchNorm buffers differ"
