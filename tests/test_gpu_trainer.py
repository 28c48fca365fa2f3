"""End-to-end GPU parity: the HIP-backed Trainer (networks + fused loss + flat Adam) against the CPU oracle
harness on identical weights, inputs and tie-break noise (SURVEY.md §8d 'AbsRel parity' proxy (i))."""
import os

import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_close
from oracle import trainer as OT

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def _opts(**over):
    from fusiondepth_amd.options import MonodepthOptions
    o = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "2",
                                  "--height", "64", "--width", "96"])
    for k, v in over.items():
        setattr(o, k, v)
    return o


def _make_oracle(opt, seed=3):
    """The oracle trainer in the deterministic initial state every test pair starts from."""
    oopt = OT.default_opt(height=opt.height, width=opt.width, batch_size=opt.batch_size, num_layers=opt.num_layers,
                          learning_rate=opt.learning_rate)
    omodels = OT.build_models(oopt, seed)
    for k, m in omodels.items():
        gin.fill_params(m, 100 + len(k))
    return OT.OracleTrainer(oopt, models=omodels)


def _make_pair(opt, seed=3):
    from fusiondepth_amd.trainer import Trainer
    tr = Trainer(opt, verbose=False, materialize_outputs=True)
    ot = _make_oracle(opt, seed)
    with torch.no_grad():
        for k, m in ot.models.items():
            for name, t in tr.models[k].state_dict().items():
                t.copy_(m.state_dict()[name])
    return tr, ot


def _batch(B, H, W, seed):
    """CPU batch (for the oracle) with the LiDAR 2-channel maps made by the oracle's C scatter."""
    from oracle import scatter as OS
    from fusiondepth_amd import functional as FD
    inp, rng = gin.batch_inputs(seed, B, H, W)
    roi = FD.scaled_roi(H, W)
    for i, f in enumerate((0, -1, 1)):
        beam = gin.lidar_4beam(np.random.RandomState(seed + 10 + i), B, H, W)
        # keep the returns near what an untrained net predicts (depth*26 ~ 5 m) so the SI-loss mask is never empty
        beam = np.where(beam > 0, 0.035 + (beam - 0.05) * (0.035 / 0.6), 0).astype(np.float32)
        two = np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0], roi)) for b in range(B)])
        inp[("2channel", f, 0)] = torch.from_numpy(two)
        if f == 0:
            inp["2channel"] = torch.from_numpy(two)
            inp["4beam"] = torch.from_numpy(beam)
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


def _rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float((np.abs(a - ref) / np.maximum(np.abs(ref), 1e-30)).max())


def _float64_models(models):
    import copy
    out = {k: copy.deepcopy(m).double() for k, m in models.items()}
    for m in out.values():
        m.train()
    return out


def _to64(inp):
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}


def _check_forward_against_float64(tag, outs_g, losses_g, ot, inp, noise, keys, floor=1e-4):
    """One training forward of the HIP trainer against GROUND TRUTH = the oracle's graph in float64, tensor by tensor (worst
    element-wise relative error) and loss by loss.  Bound: the north-star 1e-4, or twice the error the reference's own float32
    arithmetic (the float32 oracle, same weights) shows on that tensor - whichever is larger.  A 10x regression cannot pass: the
    measured errors sit at 0.3 - 1x the float32 oracle's.  Every measured error goes to the terminal summary (conftest.report)."""
    import conftest
    with torch.no_grad():
        o32, l32 = OT.process_batch(ot.opt, ot.models, {k: v.clone() for k, v in inp.items()}, noise)
        o64, l64 = OT.process_batch(ot.opt, _float64_models(ot.models), _to64(inp), [n.double() for n in noise])
    for key in keys:
        r64 = o64[key].numpy()
        e_hip, e_ref = _rel_err(outs_g[key].detach().cpu().numpy(), r64), _rel_err(o32[key].numpy(), r64)
        bound = max(floor, 2 * e_ref)
        conftest.report("%s %s" % (tag, key), e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s %s: HIP %.3g vs float64, float32 oracle %.3g" % (tag, key, e_hip, e_ref)
    for k in l64:
        r64 = float(l64[k])
        e_hip, e_ref = abs(float(losses_g[k]) - r64) / abs(r64), abs(float(l32[k]) - r64) / abs(r64)
        bound = max(1e-4, 2 * e_ref)
        conftest.report("%s %s" % (tag, k), e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s %s: HIP %.3g vs float64, float32 oracle %.3g" % (tag, k, e_hip, e_ref)


def test_trainer_matches_oracle_over_optimizer_steps():
    opt = _opts(height=128, width=192)     # layer4 BatchNorm then sees 2*4*6 = 48 samples per channel (64x96 gives 12)
    tr, ot = _make_pair(opt)
    assert tr.accumulate_step == ot.hp.accumulate_step == 1 and tr.batch_size == 2
    assert abs(tr.lr - ot.hp.learning_rate) < 1e-12
    B, H, W = 2, opt.height, opt.width
    traj_g, traj_o = [], []
    for step in range(3):
        inp, noise = _batch(B, H, W, 900 + step)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        outs_o, losses_o = ot.micro_step({k: v.clone() for k, v in inp.items()}, noise)
        if step == 0:   # gradient parity before the first update (BN buffers are restored after this extra forward)
            saved = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in tr.models.items()}
            outs_g, losses_g = tr.process_batch(ginp)
            # the oracle above has already stepped: the float32 / float64 yardsticks come from a fresh oracle in the pair's
            # (seeded, deterministic) initial state, which the HIP trainer still holds
            ot0 = _make_oracle(opt)
            _check_forward_against_float64("R18 128x192 b2 step0", outs_g, losses_g, ot0, inp, noise,
                                           [("disp", s) for s in range(4)] + [("depth", 0, s) for s in range(4)])
            for f in (-1, 1):      # 4x4 matrices with exact 0 / 1 entries: absolute bound on top of the relative one
                assert_close(outs_g[("cam_T_cam", 0, f)].detach().cpu().numpy(), outs_o[("cam_T_cam", 0, f)].detach().numpy(),
                             rtol=1e-4, atol=2e-6, what="cam_T_cam %d" % f)
            tr.flat.zero_grad()
            with torch.no_grad():
                for k, m in tr.models.items():
                    for n, b in m.named_buffers():
                        b.copy_(saved[k][n])
        losses_g = tr.train_step([ginp])
        traj_g.append(float(losses_g["loss"]))
        traj_o.append(float(losses_o["loss"]))
    print("loss trajectory HIP %s | oracle %s" % (traj_g, traj_o))
    assert np.isfinite(traj_g).all() and np.isfinite(traj_o).all(), "SI-loss mask became empty: fix the test inputs"
    # (the AbsRel clause of the north star has its own test: test_absrel_after_equal_steps_vs_oracle_fixture)
    assert_close(traj_g[0], traj_o[0], rtol=1e-4, atol=0, what="loss at step 0")
    # later steps depend on Adam updates of ~49M parameters; the first updates are ~lr*sign(g), so noise-level gradients
    # (|g| ~ 1e-8) move parameters by +-lr depending on rounding: the trajectories separate at the 1e-4..1e-3 level
    assert_close(traj_g, traj_o, rtol=5e-3, atol=0, what="loss trajectory")
    # parameters after 3 steps: Adam moves each weight by <= lr per step; compare the flat buffers' statistics
    po = torch.cat([p.detach().reshape(-1) for p in OT.trainable_parameters(ot.models)])
    pg = tr.flat.flat_param.cpu()
    assert pg.numel() == po.numel()
    assert float((pg - po).abs().max()) <= 3 * 3 * tr.lr + 1e-6, "a parameter moved further than Adam allows"


@pytest.mark.parametrize("layers,H,W,B", [(50, 64, 96, 2), (18, 320, 1024, 2), (18, 352, 1216, 1)],
                         ids=["C3-resnet50", "C4-1024x320", "completor-1216x352"])
def test_other_baseline_configs_match_oracle(layers, H, W, B):
    """BASELINE.json configs 3 (ResNet-50) and 4 (1024x320), and the completor's 1216x352 resolution (completor.py:31-34;
    widths / heights that are not powers of two times the tile sizes), as parity cases: losses and disparities of the first forward
    pass, then the loss after one optimiser step, against the oracle harness (same weights, inputs, tie-break noise)."""
    opt = _opts(num_layers=layers, height=H, width=W, batch_size=B)
    tr, ot = _make_pair(opt)
    # second yardstick for the small ResNet-50 case (2x3-pixel layer4, 12 samples per BatchNorm channel: one Adam step amplifies
    # rounding to the percent level): the same oracle stepped in float64 from the same state
    ot64 = None
    if layers == 50:
        ot64 = OT.OracleTrainer(ot.opt, models=_float64_models(ot.models))
    losses_seq, losses_64 = [], []
    for step in range(2):
        inp, noise = _batch(B, H, W, 700 + step)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        if ot64 is not None:
            losses_64.append(float(ot64.micro_step(_to64(inp), [n.double() for n in noise])[1]["loss"]))
        outs_o, losses_o = ot.micro_step({k: v.clone() for k, v in inp.items()}, noise)
        if step == 0:
            saved = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in tr.models.items()}
            outs_g, losses_g = tr.process_batch(ginp)
            ot0 = _make_oracle(opt)          # same initial weights as the pair above (the oracle has already stepped)
            # (round 3: this bound caught the BatchNorm statistics shift - 1.1e-4 / 1.9e-4 at 1024x320 and 1216x352 before the
            # median-of-9 shift of norm.hip, 3e-5 = the float32 oracle's own error after it)
            _check_forward_against_float64("R%d %dx%d b%d step0" % (layers, W, H, B), outs_g, losses_g, ot0, inp, noise,
                                           [("disp", s) for s in range(4)])
            tr.flat.zero_grad()
            with torch.no_grad():
                for k, m in tr.models.items():
                    for n, b in m.named_buffers():
                        b.copy_(saved[k][n])
        lg = tr.train_step([ginp])
        losses_seq.append((float(lg["loss"]), float(losses_o["loss"])))
    print("loss (HIP, oracle) per step:", losses_seq)
    assert np.isfinite(losses_seq).all()
    assert_close(losses_seq[0][0], losses_seq[0][1], rtol=5e-4, atol=0, what="loss at step 0")
    tol = 1e-2
    if losses_64:
        own = abs(losses_seq[1][1] - losses_64[1]) / losses_64[1]
        print("float32 oracle vs float64 oracle after one step: %.2e; HIP vs float64: %.2e" % (own, abs(losses_seq[1][0] - losses_64[1]) / losses_64[1]))
        tol = max(tol, 3 * own)
        assert abs(losses_seq[1][0] - losses_64[1]) <= tol * losses_64[1], "loss after one optimiser step vs the float64 oracle"
    else:
        assert_close(losses_seq[1][0], losses_seq[1][1], rtol=tol, atol=0, what="loss after one optimiser step")


def test_absrel_after_equal_steps_vs_oracle_fixture(golden):
    """The north star's AbsRel clause (BASELINE.json: "AbsRel within 0.001 of the reference after equal steps"; metric =
    trainer.py:598-630 / evaluate_depth.py:42-60): 20 optimiser steps of the HIP trainer and of the CPU oracle from the same initial
    state on the same scene batches (tests/golden/make_absrel.py: a consistent synthetic scene with a ground-truth depth field, so
    AbsRel lies in a meaningful range - 0.31 .. 0.60 - instead of ~1 against random ground truth), AbsRel of the held-out scenes
    after 0, 2, .. 20 steps.  The fixture holds the oracle's float32 AND float64 runs: float64 is ground truth, |float32 - float64|
    is what the reference's own arithmetic drifts by (from-scratch training is chaotic: 4e-7 after 2 steps, 2e-4 after 4, 4e-3
    after 6, 1e-2 after 20 - beyond step 4 the reference cannot hold 0.001 against ITSELF in float64), and nine more float32 runs
    of the oracle whose initial weights were each moved by ONE ulp (three over all 20 steps, six over the first 8): after 2 steps
    they sit within 1.5e-6 of the unperturbed run, after 4 steps 0.2 - 1.8e-3 away (all nine on the same side: the unperturbed
    run is the lowest of the ten), after 6 steps up to 4e-3.  Bound at every checkpoint: |AbsRel_HIP - AbsRel_oracle32| <= 0.001
    absolute, or three times the spread of the oracle's own runs {float64, one-ulp} around its float32 run where that exceeds
    0.001.  The HIP trainer differs from the oracle by the rounding of every kernel in every step, not by one ulp once: 4e-6
    after 2 steps (a few times the one-ulp runs' 1.5e-6, asserted <= 5e-5 - the checkpoint where 0.001 means something), and
    1.3e-3 .. 2.5e-3 after 4 steps depending on the build (any change of summation order moves it: the same amplification, from
    a few times the one-ulp perturbation).  All values go to the terminal summary."""
    import conftest
    import make_absrel as MA
    g = golden("absrel_r18_192x640_b2")
    opt = _opts(height=MA.H, width=MA.W, batch_size=MA.B, learning_rate=MA.LR)
    tr, _ = _make_pair(opt)
    assert abs(tr.lr - 2.5e-5) < 1e-12 and tr.accumulate_step == 1 and int(g["steps"]) == MA.STEPS      # 1e-4 * (batch 2 / 8), trainer.py:38
    val = []
    for seed in MA.VAL_SEEDS:
        inp, _ = MA.scene_batch(seed)
        val.append({k: v.cuda() for k, v in inp.items()})
    got = [float(tr.val_metrics(val)["de/abs_rel"])]
    losses = []
    for step in range(MA.STEPS):
        inp, noise = MA.scene_batch(MA.TRAIN_SEED + step)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        losses.append(float(tr.train_step([ginp])["loss"]))
        if (step + 1) % MA.EVERY == 0:
            got.append(float(tr.val_metrics(val)["de/abs_rel"]))
    f32, f64 = g["f32/metrics"][:, 0], g["f64/metrics"][:, 0]
    assert np.isfinite(losses).all() and np.isfinite(got).all()
    assert_close(losses[0], g["f32/loss"][0], rtol=1e-4, atol=0, what="loss of step 0")
    assert 0.03 < f64.min() and f64.max() < 1.0, "fixture AbsRel out of the meaningful range: %s" % f64
    others = [f64] + [g["f32p%d/metrics" % k][:, 0] for k in range(int(g["ensemble"]))]     # float64 + the one-ulp float32 runs
    others += [g["f32s%d/metrics" % k][:, 0] for k in range(int(g["ensemble_short"]))]      # (the short ones end after MA.SHORT_STEPS)
    for i, (a, r32) in enumerate(zip(got, f32)):
        spread = max(abs(float(o[i]) - r32) for o in others if len(o) > i)
        bound = max(1e-3, 3 * spread)
        conftest.report("AbsRel after %2d steps: HIP %.5f, oracle f32 %.5f, f64 %.5f; |HIP - f32|" % (i * MA.EVERY, a, r32, f64[i]),
                        abs(a - r32), bound, "(spread of the oracle's own runs %.1e)" % spread)
        assert abs(a - r32) <= bound, "AbsRel after %d steps: HIP %.5f vs oracle %.5f (float64 %.5f)" % (i * MA.EVERY, a, r32, f64[i])
    assert abs(got[0] - f32[0]) <= 1e-6, "AbsRel of the initial state (no optimiser step in between)"
    assert abs(got[1] - f32[1]) <= 5e-5, "AbsRel after 2 steps: before the chaotic amplification sets in, far inside 0.001"


def test_resnet50_full_size_forward_against_float64():
    """BASELINE.json config 3 at its real size: ResNet-50 encoders (bottleneck blocks, 2048-channel 1x1 convolutions at 6x20,
    256 -> 64 squeezes at 48x160; networks/resnet_encoder.py:62-74, 89-90), 640x192, batch 8 - one full training forward, every
    ("disp", s) and every loss against the oracle's graph in float64, bound = 1e-4 or twice the float32 oracle's own error."""
    opt = _opts(num_layers=50, height=192, width=640, batch_size=8)
    tr, ot = _make_pair(opt)
    assert tr.batch_size == 8 and tr.accumulate_step == 1
    assert list(tr.models["encoder"].num_ch_enc) == [64, 256, 512, 1024, 2048]
    inp, noise = _batch(8, 192, 640, 750)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    with torch.no_grad():
        outs_g, losses_g = tr.process_batch(ginp)
    assert np.isfinite(float(losses_g["loss"]))
    _check_forward_against_float64("R50 640x192 b8 (BASELINE config 3)", outs_g, losses_g, ot, inp, noise,
                                   [("disp", s) for s in range(4)] + [("depth", 0, 0)])


def _oracle_backward(models, opt, inp, noise, double):
    """(flat parameter gradient per network, disp gradients) of one training forward + backward of the oracle's graph (``models``
    already in the precision asked for)."""
    for m in models.values():
        for p_ in m.parameters():
            p_.grad = None
    i = _to64(inp) if double else {k: v.clone() for k, v in inp.items()}
    n = [x.double() for x in noise] if double else noise
    outs, losses = OT.process_batch(opt, models, i, n)
    for s_ in range(4):
        outs[("disp", s_)].retain_grad()
    losses["loss"].backward()
    per_net = {}
    for k, m in models.items():
        gs = [p_.grad.reshape(-1).double() if p_.grad is not None else torch.zeros(p_.numel(), dtype=torch.float64) for p_ in m.parameters()]
        per_net[k] = torch.cat(gs).numpy()
    disp = {s_: outs[("disp", s_)].grad.double().numpy() for s_ in range(4)}
    return per_net, disp, float(losses["loss"])


def test_resnet50_full_size_backward_against_float64():
    """BASELINE.json config 3 at its real size, BACKWARD: ResNet-50 encoders, 640x192, batch 8 - the
    gradient of the training loss w.r.t. every ("disp", s) and w.r.t. the parameters of every network (flat, per network: relative L2
    distance and the gradient norm) against the oracle's graph in float64 (reference: networks/resnet_encoder.py:62-74, trainer.py:
    268-319, 425-596).  Yardstick as everywhere: the float32 oracle's own distance from float64 on the same quantity - the bound is
    twice that (three times for the norms), with a floor of 1e-4 (norms) / 1e-3 (relative L2 of a 24-million-entry gradient,
    disp-gradient maps: pixels within rounding of an argmin / clamp tie take either branch in any float32 evaluation).  Measured: the
    float32 oracle's parameter gradients are 2.5 - 4.4 % (relative L2) from float64 at this size - train-mode BatchNorm over 8
    images - and the HIP path's 2.4 - 4.9 %."""
    import conftest
    opt = _opts(num_layers=50, height=192, width=640, batch_size=8)
    tr, ot = _make_pair(opt)
    assert tr.batch_size == 8 and tr.accumulate_step == 1
    inp, noise = _batch(8, 192, 640, 751)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    tr.flat.zero_grad()
    outs_g, losses_g = tr.process_batch(ginp)
    for s_ in range(4):
        outs_g[("disp", s_)].retain_grad()
    losses_g["loss"].backward()
    tr._join_side_streams()
    torch.cuda.synchronize()
    hip_net, off = {}, 0
    flat = tr.flat.flat_grad.double().cpu().numpy()
    for k, m in tr.models.items():
        n = sum(p_.numel() for p_ in m.parameters())
        hip_net[k] = flat[off:off + n]
        off += n
    assert off == flat.size
    hip_disp = {s_: outs_g[("disp", s_)].grad.double().cpu().numpy() for s_ in range(4)}
    m64 = _float64_models(ot.models)          # (copied before the float32 pass leaves non-leaf tensors on the modules)
    n32, d32, l32 = _oracle_backward(ot.models, ot.opt, inp, noise, False)
    n64, d64, l64 = _oracle_backward(m64, ot.opt, inp, noise, True)
    assert abs(float(losses_g["loss"]) - l64) <= 1e-4 * abs(l64)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    for k in n64:
        assert hip_net[k].shape == n64[k].shape, k
        if k in ("encoder", "beam_encoder", "beam_encoder_pose", "pose_encoder"):
            assert np.linalg.norm(n64[k]) > 0
        e_hip, e_ref = rel(hip_net[k], n64[k]), rel(n32[k], n64[k])
        bound = max(1e-3, 2 * e_ref)
        conftest.report("R50 640x192 b8 backward: %s parameter gradient, relative L2 vs float64" % k, e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s: HIP %.3g, float32 oracle %.3g" % (k, e_hip, e_ref)
        g64 = float(np.linalg.norm(n64[k]))
        e_hip, e_ref = abs(float(np.linalg.norm(hip_net[k])) - g64) / g64, abs(float(np.linalg.norm(n32[k])) - g64) / g64
        # (a norm is ONE scalar per network: where the float32 oracle lands inside its own error band is a coin toss - three times
        # its error, like the per-tensor bounds of test_resnet_encoder_fwd_bwd_vs_oracle; measured 0.6 - 2.4x)
        bound = max(1e-4, 3 * e_ref)
        conftest.report("R50 640x192 b8 backward: %s gradient norm" % k, e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s norm: HIP %.3g, float32 oracle %.3g" % (k, e_hip, e_ref)
    for s_ in range(4):
        l1 = lambda a, b: float(np.abs(a - b).sum() / np.abs(b).sum())
        e_hip, e_ref = l1(hip_disp[s_], d64[s_]), l1(d32[s_], d64[s_])
        bound = max(1e-3, 2 * e_ref)
        conftest.report("R50 640x192 b8 backward: d loss / d disp scale %d, relative L1 vs float64" % s_, e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "disp grad %d: HIP %.3g, float32 oracle %.3g" % (s_, e_hip, e_ref)


def _oracle_backward_accumulated(models, opt, mbs, double):
    """The reference's accumulate-then-step window (trainer.py:237-248): per micro-batch ``(loss / accumulate_step).backward()``,
    parameter gradients summed by autograd.  -> (flat gradient per network, [per micro-batch {scale: d window-loss / d disp}], window loss)"""
    for m in models.values():
        for p_ in m.parameters():
            p_.grad = None
    disp, total = [], 0.0
    for inp, noise in mbs:
        i = _to64(inp) if double else {k: v.clone() for k, v in inp.items()}
        n = [x.double() for x in noise] if double else noise
        outs, losses = OT.process_batch(opt, models, i, n)
        for s_ in range(4):
            outs[("disp", s_)].retain_grad()
        (losses["loss"] / len(mbs)).backward()
        total += float(losses["loss"]) / len(mbs)
        disp.append({s_: outs[("disp", s_)].grad.double().numpy() for s_ in range(4)})
    per_net = {}
    for k, m in models.items():
        gs = [p_.grad.reshape(-1).double() if p_.grad is not None else torch.zeros(p_.numel(), dtype=torch.float64) for p_ in m.parameters()]
        per_net[k] = torch.cat(gs).numpy()
    return per_net, disp, total


def _full_size_backward_against_float64(tag, H, W, B, groups, seed):
    """BACKWARD of one optimiser step's window at a BASELINE configuration's real size ("missing" 3): the gradient of
    the window loss w.r.t. every parameter of every network (flat per network: relative L2 distance and norm) and w.r.t. every
    ("disp", s), of the HIP trainer's ONE stacked pass over the ``groups`` micro-batches (grouped BatchNorm statistics, per-group loss
    reductions: trainer.py:237-248, 268-319, 425-596) against ``groups`` separate passes of the oracle's graph in float64, summed.
    Yardstick: the float32 oracle's own distance from float64 on the same quantity; bounds as in the ResNet-50 test
    (max(1e-3, 2x) on relative L2 / L1 - 3x for the depth decoder, see below -, max(1e-4, 3x, the float32 oracle's relative L2) on
    the norms)."""
    import conftest
    opt = _opts(num_layers=18, height=H, width=W, batch_size=B * groups)
    tr, ot = _make_pair(opt)
    assert tr.batch_size == B and tr.accumulate_step == groups
    mbs = [_batch(B, H, W, seed + g) for g in range(groups)]
    ginps = []
    for inp, noise in mbs:
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        ginps.append(g)
    tr.flat.zero_grad()
    if groups == 1:
        outs_g, losses_g = tr.process_batch(ginps[0])
    else:
        outs_g, losses_g = tr.process_batch(tr.stack_micro_batches(ginps), groups=groups)
    for s_ in range(4):
        outs_g[("disp", s_)].retain_grad()
    losses_g["loss"].backward()
    tr._join_side_streams()
    torch.cuda.synchronize()
    hip_net, off = {}, 0
    flat = tr.flat.flat_grad.double().cpu().numpy()
    for k, m in tr.models.items():
        n = sum(p_.numel() for p_ in m.parameters())
        hip_net[k] = flat[off:off + n]
        off += n
    assert off == flat.size
    hip_disp = {s_: outs_g[("disp", s_)].grad.double().cpu().numpy() for s_ in range(4)}
    m64 = _float64_models(ot.models)          # (copied before the float32 pass leaves non-leaf tensors on the modules)
    n32, d32, l32 = _oracle_backward_accumulated(ot.models, ot.opt, mbs, False)
    n64, d64, l64 = _oracle_backward_accumulated(m64, ot.opt, mbs, True)
    assert abs(float(losses_g["loss"]) - l64) <= 1e-4 * abs(l64)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    for k in n64:
        assert hip_net[k].shape == n64[k].shape, k
        assert np.linalg.norm(n64[k]) > 0, k
        e_hip, e_ref = rel(hip_net[k], n64[k]), rel(n32[k], n64[k])
        e_ref_l2 = e_ref
        # The depth decoder's gradient is the linear image of the d loss / d disp maps, whose pixels within rounding of an argmin /
        # clamp tie take either branch in any float32 evaluation (the documented deviation of the loss path): errors that do not
        # average out like rounding noise.  Measured 2.0x the float32 oracle's own error at both sizes, with the limb GEMMs on
        # (2.495e-3) and off (2.462e-3) alike - the pose networks' last bit decides which pixels flip; bound 3x for that network.
        bound = max(1e-3, (3 if k == "depth" else 2) * e_ref)
        conftest.report("%s backward: %s parameter gradient, relative L2 vs float64" % (tag, k), e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s: HIP %.3g, float32 oracle %.3g" % (k, e_hip, e_ref)
        g64 = float(np.linalg.norm(n64[k]))
        e_hip, e_ref = abs(float(np.linalg.norm(hip_net[k])) - g64) / g64, abs(float(np.linalg.norm(n32[k])) - g64) / g64
        # (one scalar per network: where the float32 oracle's norm lands inside its own error band is a coin toss - at 1024x320 its
        # depth-decoder norm is 1e-6 from float64 while the vector is 1.7e-3 away - so the bound never drops below that vector error)
        bound = max(1e-4, 3 * e_ref, e_ref_l2)
        conftest.report("%s backward: %s gradient norm" % (tag, k), e_hip, bound, "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= bound, "%s norm: HIP %.3g, float32 oracle %.3g" % (k, e_hip, e_ref)
    l1 = lambda a, b: float(np.abs(a - b).sum() / np.abs(b).sum())
    for g in range(groups):
        for s_ in range(4):
            got = hip_disp[s_][g * B:(g + 1) * B]
            e_hip, e_ref = l1(got, d64[g][s_]), l1(d32[g][s_], d64[g][s_])
            bound = max(1e-3, 2 * e_ref)
            conftest.report("%s backward: micro-batch %d d loss / d disp scale %d, relative L1 vs float64" % (tag, g, s_), e_hip, bound,
                            "(float32 oracle %.2e)" % e_ref)
            assert e_hip <= bound, "micro-batch %d disp grad %d: HIP %.3g, float32 oracle %.3g" % (g, s_, e_hip, e_ref)


def test_resnet18_batch12_stacked_backward_against_float64():
    """BASELINE.json config 2 exactly - what bench.py times: ResNet-18, 640x192, --batch_size 12 = 2 micro-batches of 6 as ONE stacked
    pass; parameter gradients of all six networks and d loss / d disp_s against two float64 oracle passes summed."""
    _full_size_backward_against_float64("R18 640x192 b6x2 (BASELINE config 2)", 192, 640, 6, 2, 770)


def test_resnet18_1024x320_batch8_backward_against_float64():
    """One rank's share of BASELINE.json config 4 at its real size, backward: ResNet-18, 1024x320, --batch_size 8."""
    _full_size_backward_against_float64("R18 1024x320 b8 (BASELINE config 4, one rank)", 320, 1024, 8, 1, 775)


def test_resnet18_1024x320_batch8_forward_against_float64():
    """One rank's share of BASELINE.json config 4 at its real size: ResNet-18, 1024x320, --batch_size 8
    (one micro-batch of 8) - every ("disp", s), ("depth", 0, 0) and every loss of a training forward against float64."""
    opt = _opts(num_layers=18, height=320, width=1024, batch_size=8)
    tr, ot = _make_pair(opt)
    assert tr.batch_size == 8 and tr.accumulate_step == 1
    inp, noise = _batch(8, 320, 1024, 760)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    with torch.no_grad():
        outs_g, losses_g = tr.process_batch(ginp)
    assert np.isfinite(float(losses_g["loss"]))
    _check_forward_against_float64("R18 1024x320 b8 (BASELINE config 4, one rank)", outs_g, losses_g, ot, inp, noise,
                                   [("disp", s) for s in range(4)] + [("depth", 0, 0)])


_ABSREL_RUNS = {}


def _hip_absrel_runs(n_streams):
    """AbsRel of the two held-out scenes after 0 / 20 / 40 / 60 optimiser steps of the HIP trainer on data streams 0 .. n_streams-1 of
    tests/golden/make_absrel_stat.py (same initial state, same batches, same tie-break noise as the oracle runs of the fixtures);
    computed once per session, shared by the distribution test (streams 0-5) and the paired test (streams 0-11)."""
    import make_absrel_stat as MS
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.trainer import Trainer
    have = _ABSREL_RUNS.get("got")
    if have is not None and have.shape[0] >= n_streams:
        return have[:n_streams]
    val = []
    for seed in MS.VAL_SEEDS:
        inp, _ = MS.scene_batch(seed)
        val.append({k: v.cuda() for k, v in inp.items()})
    got = np.zeros((n_streams, len(MS.CHECK)))
    for k in range(n_streams):
        opt = _opts(height=MS.H, width=MS.W, batch_size=MS.B, learning_rate=MS.LR)
        tr = Trainer(opt, verbose=False)
        om = MS.models(MS.oracle_opt(), k)
        with torch.no_grad():
            for name, m in om.items():
                for n_, t in tr.models[name].state_dict().items():
                    t.copy_(m.state_dict()[n_])
        FD.bump_weights_epoch()
        assert abs(tr.lr - 2.5e-5) < 1e-12 and tr.accumulate_step == 1
        got[k, 0] = float(tr.val_metrics(val)["de/abs_rel"])
        ci = 1
        for step in range(MS.STEPS):
            inp, noise = MS.scene_batch(MS.train_seed(k, step))
            ginp = {kk: v.cuda() for kk, v in inp.items()}
            ginp["_noise"] = [n.cuda() for n in noise]
            tr.train_step([ginp])
            if (step + 1) in MS.CHECK:
                got[k, ci] = float(tr.val_metrics(val)["de/abs_rel"])
                ci += 1
        del tr
    assert np.isfinite(got).all()
    _ABSREL_RUNS["got"] = got
    return got


def test_absrel_distribution_matches_the_oracle(golden):
    """The long-run form of the north star's AbsRel clause (trainer.py:598-630, layers.py:284-302).
    Trajectory-wise agreement ends after ~4 optimiser steps for ANY two float32 implementations
    (test_absrel_after_equal_steps_vs_oracle_fixture); over 60 steps the claim that can be true - and is tested here - is
    statistical: K = 6 independent runs (own stream of scene batches) of the HIP trainer and of the CPU oracle
    (tests/golden/make_absrel_stat.py, generated in the build container) give AbsRel samples after 20, 40 and 60 steps
    whose means differ by no more than max(0.001, 2 standard errors of the difference) and whose ranges overlap; the starting
    points (0 steps) agree run by run to 1e-5.  (A sanity check; the sharper, paired comparison is the next test.)"""
    import conftest
    import make_absrel_stat as MS
    g = golden(MS.NAME)
    want = g["metrics"][:, :, 0]                                   # [K, checkpoints]
    assert want.shape == (MS.K, len(MS.CHECK)) and int(g["steps"]) == MS.STEPS
    got = _hip_absrel_runs(MS.K)
    assert np.abs(got[:, 0] - want[:, 0]).max() <= 1e-5, "initial states differ: %s vs %s" % (got[:, 0], want[:, 0])
    for ci in range(1, len(MS.CHECK)):
        h, o = got[:, ci], want[:, ci]
        se = float(np.sqrt(h.var(ddof=1) / MS.K + o.var(ddof=1) / MS.K))
        diff = abs(float(h.mean() - o.mean()))
        bound = max(1e-3, 2 * se)
        conftest.report("AbsRel after %2d steps, %d runs each: HIP mean %.5f (%.5f .. %.5f), oracle mean %.5f (%.5f .. %.5f); |difference of means|"
                        % (MS.CHECK[ci], MS.K, h.mean(), h.min(), h.max(), o.mean(), o.min(), o.max()), diff, bound, "(2 SE = %.1e)" % (2 * se))
        assert diff <= bound, "AbsRel after %d steps: HIP %s vs oracle %s" % (MS.CHECK[ci], h, o)
        assert h.min() <= o.max() and o.min() <= h.max(), "AbsRel ranges after %d steps do not overlap: HIP %s, oracle %s" % (MS.CHECK[ci], h, o)


def test_absrel_paired_gap_vs_oracle(golden):
    """the PAIRED design.  Twelve data streams; on each, the HIP trainer, the float32 oracle ("base") and the
    float32 oracle started one ulp away ("ulp") run 60 optimiser steps from one initial state over exactly the same batches and noise
    (tests/golden/make_absrel_paired.py).  The statistic is the per-stream difference HIP - base of the held-out AbsRel at steps
    20 / 40 / 60: its mean over the streams must be within max(0.001, 2 standard errors of that mean) of zero; the oracle's own paired
    gap ulp - base - what two equally valid float32 implementations of the reference differ by - is printed beside it.  What pairing
    does NOT do here is shrink the spread: the fixture shows the one-ulp partner of a stream as far from it after 20 steps (SE of the
    paired gap 0.04 - 0.06) as another stream is (unpaired SE 0.04 - 0.06), i.e. the training dynamics have forgotten the common
    start - which is the reason the 0.001 of the north star can only be shown for the first steps
    (test_absrel_after_equal_steps_vs_oracle_fixture: 4e-6 after 2 steps)."""
    import conftest
    import make_absrel_paired as MP
    import make_absrel_stat as MS
    g = golden(MP.NAME)
    base, ulp = g["base"][:, :, 0], g["ulp"][:, :, 0]              # [K, checkpoints]
    K = int(g["streams"])
    assert base.shape == (K, len(MS.CHECK)) and K == MP.K and K >= 12
    got = _hip_absrel_runs(K)
    assert np.abs(got[:, 0] - base[:, 0]).max() <= 1e-5, "initial states differ: %s vs %s" % (got[:, 0], base[:, 0])
    thr = golden(MS.NAME)["metrics"][:, :, 0] - base[:MS.K]        # the SAME oracle on the same streams, generated on 16 threads instead of 4
    for ci in range(1, len(MS.CHECK)):
        conftest.report("AbsRel after %2d steps: oracle on 16 CPU threads - oracle on 4 threads, streams 0-5 = %+.4f +- %.4f; largest |gap|"
                        % (MS.CHECK[ci], thr[:, ci].mean(), thr[:, ci].std(ddof=1) / np.sqrt(MS.K)), np.abs(thr[:, ci]).max(), float("inf"), "(information)")
    for ci in range(1, len(MS.CHECK)):
        d_hip, d_ulp = got[:, ci] - base[:, ci], ulp[:, ci] - base[:, ci]
        mean, se = float(d_hip.mean()), float(d_hip.std(ddof=1) / np.sqrt(K))
        mean_o, se_o = float(d_ulp.mean()), float(d_ulp.std(ddof=1) / np.sqrt(K))
        bound = max(1e-3, 2 * se)
        conftest.report("AbsRel after %2d steps, %d paired streams: HIP - oracle = %+.4f +- %.4f (SE); oracle(one ulp away) - oracle = %+.4f +- %.4f; |mean paired gap|"
                        % (MS.CHECK[ci], K, mean, se, mean_o, se_o), abs(mean), bound, "(2 SE = %.1e)" % (2 * se))
        assert abs(mean) <= bound, "paired AbsRel gap after %d steps: %+.4f +- %.4f (per stream: %s)" % (MS.CHECK[ci], mean, se, d_hip)
        # and the HIP gap is not an outlier against what the oracle differs from itself by: same order of spread
        assert se <= 3 * se_o + 1e-3, "paired spread after %d steps: HIP %.4f vs oracle-vs-itself %.4f" % (MS.CHECK[ci], se, se_o)


def test_depth_monitoring_metrics_vs_reference_golden(golden):
    """Trainer.compute_depth_losses (device-side resize / crop / median scaling / metrics kernel) against the metrics the
    reference's own Trainer.compute_depth_losses produced on the same inputs (tests/golden/make_golden.py)."""
    g = golden("depth_losses_b2_192x640")
    gt, pred = gin.depth_eval_inputs(int(g["seed"]), 2, 192, 640)
    opt = _opts(height=192, width=640)
    from fusiondepth_amd.trainer import Trainer
    tr = Trainer(opt, verbose=False)
    losses = {}
    tr.compute_depth_losses({"depth_gt": torch.from_numpy(gt).cuda()}, {("depth", 0, 0): torch.from_numpy(pred).cuda()}, losses)
    got = [float(losses[m]) for m in tr.depth_metric_names]
    assert_close(got, g["metrics"], rtol=2e-5, atol=0, what="depth metrics vs reference")


def test_accumulate_semantics_batch12():
    """--batch_size 12 -> accumulate 2 x micro-batch 6, lr 1.5e-4, StepLR step 6 (trainer.py:28-41)."""
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import synthetic
    opt = _opts(batch_size=12)
    tr = Trainer(opt, verbose=False)
    assert (tr.accumulate_step, tr.batch_size, tr.scheduler_step_size, tr.opt.num_epochs) == (2, 6, 6, 11)
    assert abs(tr.lr - 1.5e-4) < 1e-12
    mbs = [synthetic.make_batch(6, 64, 96, seed=s) for s in (1, 2)]
    p0 = tr.flat.flat_param.clone()
    losses = tr.train_step(mbs)
    assert torch.isfinite(losses["loss"]).item()
    assert tr.adam_step_count == 1 and tr.batch_idx == 2
    moved = (tr.flat.flat_param - p0).abs()
    assert float(moved.max()) <= 1.5e-4 * 1.001 and float(moved.max()) > 0
    assert float(tr.flat.flat_grad.abs().max()) == 0.0, "zero_grad after the step"
    for _ in range(6):
        tr.end_epoch()
    assert abs(tr.lr - 1.5e-5) < 1e-12


def test_checkpoint_layout_roundtrip(tmp_path):
    from fusiondepth_amd.trainer import Trainer
    opt = _opts(height=32, width=64, log_dir=str(tmp_path))
    tr = Trainer(opt, verbose=False)
    folder = tr.save_model("best")
    files = sorted(p.name for p in (tmp_path / "mdp" / "models" / "weights_best").iterdir())
    assert files == sorted(["adam.pth", "beam_encoder.pth", "beam_encoder_pose.pth", "depth.pth", "encoder.pth",
                            "pose.pth", "pose_encoder.pth"])
    enc = torch.load(folder + "/encoder.pth")
    assert enc["height"] == 32 and enc["width"] == 64 and "encoder.conv1.weight" in enc and "encoder.fc.weight" in enc
    opt2 = _opts(height=32, width=64, log_dir=str(tmp_path), train_load_weights_folder=folder,
                 models_to_load=["encoder", "depth", "pose_encoder", "pose", "beam_encoder", "beam_encoder_pose"])
    tr2 = Trainer(opt2, verbose=False)
    assert torch.equal(tr2.flat.flat_param, tr.flat.flat_param)


def test_graph_replay_matches_eager():
    """A captured hipGraph of the optimiser step must produce the same parameters as eager launches."""
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import synthetic
    outs = []
    for graphed in (False, True):
        torch.manual_seed(0)
        tr = Trainer(_opts(), verbose=False)
        with torch.no_grad():
            tr.flat.flat_param.copy_(torch.linspace(-0.05, 0.05, tr.flat.numel(), device="cuda").sin() * 0.05)
        mb = synthetic.make_batch(2, 64, 96, seed=5)
        mb["_noise"] = [torch.zeros(2, 2, 64, 96, device="cuda") for _ in range(4)]
        fn = tr.train_step_graphed if graphed else tr.train_step
        for _ in range(4):
            losses = fn([mb])
        torch.cuda.synchronize()
        outs.append((tr.flat.flat_param.clone(), float(losses["loss"]), float(tr.adam_state[0])))
    assert outs[0][2] == outs[1][2] == 4.0
    assert_close(outs[1][1], outs[0][1], rtol=1e-5, atol=0, what="loss after 4 steps, graph vs eager")
    assert_close(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy(), rtol=1e-4, atol=1e-6, what="parameters, graph vs eager")


def test_posecnn_variant_runs_and_matches_oracle_losses():
    """--pose_model_type posecnn (trainer.py:110-112, 352-353, 450-460): pose from PoseCNN on the image pair, translation
    rescaled by the mean inverse depth per scale."""
    from fusiondepth_amd.trainer import Trainer
    from oracle import layers as OL, networks as ON
    import torch.nn.functional as F
    opt = _opts(pose_model_type="posecnn", beam_encoder=True)
    tr = Trainer(opt, verbose=False, materialize_outputs=True)
    assert "pose_encoder" not in tr.models and type(tr.models["pose"]).__name__ == "PoseCNN"
    B, H, W = 2, 64, 96
    inp, noise = _batch(B, H, W, 321)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    outs, losses = tr.process_batch(ginp)
    assert torch.isfinite(losses["loss"]).item()
    losses["loss"].backward()
    # oracle: same weights, same wiring
    oopt = OT.default_opt(height=H, width=W)
    om = {"encoder": ON.ResnetEncoder(18, False), "beam_encoder": ON.ResnetEncoder(18, False, beam_encoder=True),
          "depth": ON.DepthDecoder(np.array([64, 64, 128, 256, 512])), "pose": ON.PoseCNN(2)}
    for k, m in om.items():
        m.load_state_dict({n: t.detach().cpu() for n, t in tr.models[k].state_dict().items()})
        m.train()
    feats = om["encoder"](inp[("color_aug", 0, 0)])
    o = dict(om["depth"](feats, beam_features=om["beam_encoder"](inp["2channel"])))
    total = 0
    for s in range(4):
        disp = F.interpolate(o[("disp", s)], [H, W], mode="bilinear", align_corners=False)
        sd, depth = OL.disp_to_depth(disp, 0.1, 100.0)
        mid = (1 / depth).mean(3, True).mean(2, True)
        o2 = dict(o)
        for f in (-1, 1):
            order = (f, 0) if f < 0 else (0, f)
            aa, tr_ = om["pose"](torch.cat([inp[("color_aug", i, 0)] for i in order], 1))
            o2[("cam_T_cam", 0, f)] = OL.transformation_from_parameters(aa[:, 0], tr_[:, 0] * mid[:, 0], f < 0)
        sopt = OT.default_opt(height=H, width=W, scales=[s])
        OT.generate_images_pred(sopt, inp, o2)
        ls = OT.compute_losses(sopt, inp, o2, {s: noise[s]})
        assert_close(float(losses["loss/%d" % s]), float(ls["loss/%d" % s]), rtol=2e-4, atol=1e-6, what="posecnn loss/%d" % s)


def test_stacked_microbatches_equal_sequential_accumulation():
    """train_step with the accumulated micro-batches stacked into one pass (grouped BatchNorm, per-micro-batch SI loss)
    must give the gradients, BatchNorm running statistics and loss terms of the reference's sequential accumulation."""
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import synthetic
    B, H, W = 5, 64, 96
    mbs = []
    for i in range(2):
        inp, noise = _batch(B, H, W, 700 + i)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        mbs.append(g)
    res = {}
    for stacked in (False, True):
        tr = Trainer(_opts(batch_size=10), verbose=False)
        assert tr.accumulate_step == 2 and tr.batch_size == 5
        with torch.no_grad():
            tr.flat.flat_param.copy_(torch.linspace(-3, 3, tr.flat.numel(), device="cuda").sin() * 0.04)
            for m in tr.models.values():
                for n, b in m.named_buffers():
                    if "running_var" in n:
                        b.fill_(1.0)
                    elif "running_mean" in n:
                        b.zero_()
        tr.stack_microbatches = stacked
        grabbed = {}
        tr.optimizer_step = lambda scale=1.0: grabbed.update(grad=tr.flat.flat_grad.clone())
        losses = tr.train_step([dict(m) for m in mbs])
        bufs = torch.cat([b.detach().float().reshape(-1) for m in tr.models.values() for n, b in m.named_buffers()])
        res[stacked] = (grabbed["grad"], bufs, {k: float(v) for k, v in losses.items()})
    g0, g1 = res[False][0].cpu().numpy().astype(np.float64), res[True][0].cpu().numpy().astype(np.float64)
    rel = np.abs(g1 - g0).sum() / np.abs(g0).sum()
    assert rel < 2e-4, "aggregate gradient difference stacked vs sequential: %.3g" % rel
    assert_close(res[True][1].cpu().numpy(), res[False][1].cpu().numpy(), rtol=1e-4, atol=1e-5, what="BN buffers after the step")
    # the sequential path reports the LAST micro-batch's losses, the stacked path the mean over micro-batches
    tr2 = Trainer(_opts(batch_size=10), verbose=False)
    assert tr2.stack_microbatches


def test_val_loop_and_best_checkpoint_bookkeeping(tmp_path):
    """trainer.py:390-423: eval-mode depth-only forward over the validation batches, mean metrics, best de/abs_rel
    remembered and saved as weights_best (+ weights_absrel<N> below 0.080)."""
    import os
    from oracle import layers as OL
    B, H, W = 1, 64, 96
    opt = _opts(batch_size=B, log_dir=str(tmp_path))
    tr, ot = _make_pair(opt)
    batches = []
    for i in range(2):
        inp, _ = _batch(B, H, W, 970 + i)
        gt, _p = gin.depth_eval_inputs(980 + i, B, H, W)               # KITTI-sized ground truth [B,1,375,1242]
        inp["depth_gt"] = torch.from_numpy(gt)
        batches.append({k: v.cuda() for k, v in inp.items()})
    losses = tr.val(batches)
    for m in ot.models.values():
        m.eval()
    want = np.zeros(7)
    with torch.no_grad():
        for b in batches:
            cb = {k: v.cpu() for k, v in b.items()}
            outs = ot.models["depth"](ot.models["encoder"](cb[("color_aug", 0, 0)]),
                                      beam_features=ot.models["beam_encoder"](cb["2channel"]))
            disp = torch.nn.functional.interpolate(outs[("disp", 0)], [H, W], mode="bilinear", align_corners=False)
            want += np.array(OT.compute_depth_losses(OL.disp_to_depth(disp, 0.1, 100.0)[1], cb["depth_gt"]))
    want /= 2
    assert_close([float(losses[n]) for n in tr.depth_metric_names], want, rtol=2e-3, atol=0, what="val metrics")
    assert tr.best == float(losses["de/abs_rel"]) < 10.0
    assert [os.path.basename(p) for p in tr.last_saved][:1] == ["weights_best"]
    assert os.path.isfile(os.path.join(tr.last_saved[0], "encoder.pth")) and os.path.isfile(os.path.join(tr.last_saved[0], "adam.pth"))
    absrel = round(float(losses["de/abs_rel"]) * 1000)
    assert (len(tr.last_saved) == 2) == (absrel < 80)
    assert all(m.training for m in tr.models.values())
    tr.val(batches)
    assert tr.last_saved == []                                          # not strictly better: nothing saved


@pytest.mark.parametrize("interleave", [False, True])
def test_gradients_are_run_to_run_identical_under_gpu_contention(interleave):
    """Every kernel is deterministic, so repeating forward + backward on the same weights must give bit-identical gradients -
    unless two streams race.  A background stream keeps the GPU busy with unrelated work of varying size to shake the timing
    (this is how a gradient tensor shared by two encoder streams was caught; see functional._UpCat.backward)."""
    B, H, W = 2, 64, 96
    opt = _opts(batch_size=B)
    from fusiondepth_amd.trainer import Trainer
    tr = Trainer(opt, verbose=False)
    tr.interleave_encoders = interleave
    inp, noise = _batch(B, H, W, 990)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    saved = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in tr.models.items()}
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (256, 1024, 2048, 3072)]
    ref = None
    for rep in range(6):
        with torch.no_grad():
            for k, m in tr.models.items():
                for n, b in m.named_buffers():
                    b.copy_(saved[k][n])
        tr.flat.zero_grad()
        torch.cuda.synchronize()
        with torch.cuda.stream(bg):
            for i in range(20 + 7 * rep):
                junk[(i + rep) % 4] @ junk[(i + rep) % 4]
        outputs, losses = tr.process_batch(ginp, groups=tr.accumulate_step)
        losses["loss"].backward()
        tr._join_side_streams()
        torch.cuda.synchronize()
        g = tr.flat.flat_grad.clone()
        if ref is None:
            ref = g
            assert float(g.abs().max()) > 0
        else:
            assert torch.equal(g, ref), "repetition %d: %d gradient entries differ" % (rep, int((g != ref).sum()))


def test_pose_decoder_and_smoothness_on_side_streams_are_bit_identical(fdtune):
    """Trainer.predict_poses runs the pose decoder (forward and, through autograd, backward) on the pose encoder's stream, and
    Trainer._process_batch the smoothness terms on the beam encoder's stream, instead of on the main stream.  Same kernels, same
    accumulation targets: the flat gradient buffer after a backward pass equals the all-on-the-main-stream result bit for bit, also
    with a background stream keeping the GPU busy (a tensor handed between streams without ordering would show up here)."""
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 2, 64, 96
    inp, noise = _batch(B, H, W, 993)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (512, 2048)]
    grads = {}
    for mode in ("0", "1"):
        fdtune.host(pose_stream=mode == "1", smooth_stream=mode == "1")
        torch.manual_seed(4321)                                  # the same initial weights for both trainers
        tr = Trainer(_opts(batch_size=B), verbose=False)
        for rep in range(3):
            tr.flat.zero_grad()
            saved = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in tr.models.items()}
            torch.cuda.synchronize()
            with torch.cuda.stream(bg):
                for i in range(8 + 7 * rep):
                    junk[i % 2] @ junk[i % 2]
            outputs, losses = tr.process_batch(ginp, groups=tr.accumulate_step)
            losses["loss"].backward()
            tr._join_side_streams()
            torch.cuda.synchronize()
            g = tr.flat.flat_grad.clone()
            with torch.no_grad():
                for k, m in tr.models.items():
                    for n, b in m.named_buffers():
                        b.copy_(saved[k][n])
            if mode in grads:
                assert torch.equal(g, grads[mode][0]), "mode %s, repetition %d" % (mode, rep)
            grads[mode] = (g, float(losses["loss"]))
    assert float(grads["0"][0].abs().max()) > 0
    assert grads["0"][1] == grads["1"][1]
    assert torch.equal(grads["0"][0], grads["1"][0]), "%d gradient entries differ" % int((grads["0"][0] != grads["1"][0]).sum())


def test_early_loss_inputs_are_bit_identical(fdtune):
    """tuning.host.early_loss_inputs: the identity reprojection losses and the tie-break noise (trainer.py:515-528, 551-552) are issued
    on an encoder's stream at the start of the step instead of between the depth decoder and the loss kernel.  Same kernels, same order
    of random draws (the noise is DRAWN here, not injected): parameters and losses over three optimiser steps equal the late-issue run
    bit for bit, with a background stream keeping the GPU busy."""
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 2, 64, 96
    batches = []
    for i in range(3):
        inp, _ = _batch(B, H, W, 1201 + i)
        batches.append({k: v.cuda() for k, v in inp.items()})
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (512, 2048)]
    res = {}
    for early in (False, True):
        fdtune.host(early_loss_inputs=early)
        torch.manual_seed(2468)
        tr = Trainer(_opts(batch_size=B), verbose=False)
        assert tr.accumulate_step == 1
        losses = []
        for rep, b in enumerate(batches):
            torch.cuda.synchronize()
            with torch.cuda.stream(bg):
                for i in range(6 + 5 * rep):
                    junk[i % 2] @ junk[i % 2]
            seen = []
            orig = tr.identity_losses
            tr.identity_losses = lambda *a, **k: (seen.append(torch.cuda.current_stream().cuda_stream), orig(*a, **k))[1]
            losses.append(float(tr.train_step([dict(b)])["loss"]))
            tr.identity_losses = orig
            main = torch.cuda.current_stream().cuda_stream
            assert len(seen) == 1 and (seen[0] != main) == early, (seen, main, early)
        torch.cuda.synchronize()
        res[early] = (tr.flat.flat_param.clone(), losses)
        del tr
    assert res[False][1] == res[True][1], (res[False][1], res[True][1])
    assert torch.isfinite(res[True][0]).all()
    assert torch.equal(res[False][0], res[True][0]), "%d parameters differ" % int((res[False][0] != res[True][0]).sum())


def test_late_weight_relayout_is_ordered_before_its_readers(fdtune):
    """functional.refresh_weight_layouts sends the big half of the post-Adam re-layout (data-gradient layouts, forward layouts from
    1 MB up) to a side stream and lets every stream that is about to use such a layout wait for it first.  Five optimiser steps with
    that side stream held back by ~50 ms of queued matrix products must leave the parameters bit-identical to the one-launch
    refresh (tuning.host.late_relayout = False): a reader that did not wait would have convolved with the previous step's weights."""
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 2, 64, 96
    batches = []
    for i in range(5):
        inp, noise = _batch(B, H, W, 1200 + i)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        batches.append(g)
    junk = torch.randn(4096, 4096, device="cuda")
    params, used = {}, 0
    for mode in ("0", "1"):
        fdtune.host(late_relayout=mode == "1")
        torch.manual_seed(4321)                                  # the same initial weights for both trainers
        tr = Trainer(_opts(batch_size=B), verbose=False)
        assert tr.accumulate_step == 1
        for g in batches:
            if mode == "1" and FD._LATE["stream"] is not None:
                with torch.cuda.stream(FD._LATE["stream"]):      # queued in front of this step's re-layout launch
                    for _ in range(40):
                        junk @ junk
            tr.train_step([g])
            used += int(mode == "1" and FD._LATE["event"] is not None)
        torch.cuda.synchronize()
        params[mode] = tr.flat.flat_param.clone()
        FD.sync_late_layouts()
        del tr
    assert used >= 3, "the side-stream re-layout never ran"
    assert torch.isfinite(params["0"]).all()
    assert torch.equal(params["0"], params["1"]), "%d parameters differ" % int((params["0"] != params["1"]).sum())


@pytest.mark.parametrize("layers", [18, 50])
def test_fused_conv_batchnorm_node_is_bit_identical(layers, fdtune):
    """functional._ConvBN (conv + BatchNorm of a ResNet block as ONE autograd node, tuning.host.fused_conv_bn) issues the same C-ABI
    calls in the same order as the two nodes it replaces: parameters (and BatchNorm running statistics) after three optimiser steps
    must be identical bit for bit, for the basic and the bottleneck block, and the fused node must really have been used."""
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 2, 64, 96
    res = {}
    for fused in (True, False):
        fdtune.host(fused_conv_bn=fused)
        n_apply = [0]
        orig = FD._ConvBN.forward

        def counting(ctx, *a, _orig=orig):
            n_apply[0] += 1
            return _orig(ctx, *a)
        FD._ConvBN.forward = staticmethod(counting)
        try:
            torch.manual_seed(2468)
            tr = Trainer(_opts(batch_size=B, num_layers=layers), verbose=False)
            for i in range(3):
                inp, noise = _batch(B, H, W, 700 + i)
                ginp = {k: v.cuda() for k, v in inp.items()}
                ginp["_noise"] = [n.cuda() for n in noise]
                tr.train_step([ginp])
            torch.cuda.synchronize()
        finally:
            FD._ConvBN.forward = staticmethod(orig)
        assert (n_apply[0] > 0) == fused, n_apply
        bufs = torch.cat([b.detach().float().flatten() for m in tr.models.values() for b in m.buffers()])
        res[fused] = (tr.flat.flat_param.clone(), bufs)
        del tr
    assert torch.isfinite(res[True][0]).all()
    assert torch.equal(res[True][0], res[False][0]), "%d parameters differ" % int((res[True][0] != res[False][0]).sum())
    assert torch.equal(res[True][1], res[False][1])


def test_fused_pose_head_is_bit_identical(fdtune):
    """tuning.host.fused_pose_head (FD.pose_head: the stacked pose network's slices, concatenations and pose matrices as one launch each
    way) against the slice-by-slice path: parameters after three optimiser steps of two accumulated micro-batches identical bit for
    bit, and the fused node really used."""
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 12, 64, 96                 # --batch_size 12 = two accumulated micro-batches of 6 (trainer.py:28-41)
    res = {}
    for fused in (True, False):
        fdtune.host(fused_pose_head=fused)
        n_apply = [0]
        orig = FD._PoseHead.forward

        def counting(ctx, *a, _orig=orig):
            n_apply[0] += 1
            return _orig(ctx, *a)
        FD._PoseHead.forward = staticmethod(counting)
        try:
            torch.manual_seed(1357)
            tr = Trainer(_opts(batch_size=B), verbose=False)
            assert tr.accumulate_step == 2
            for i in range(3):
                mbs = []
                for j in range(tr.accumulate_step):
                    inp, noise = _batch(tr.batch_size, H, W, 900 + 10 * i + j)
                    ginp = {k: v.cuda() for k, v in inp.items()}
                    ginp["_noise"] = [n.cuda() for n in noise]
                    mbs.append(ginp)
                tr.train_step(mbs)
            torch.cuda.synchronize()
        finally:
            FD._PoseHead.forward = staticmethod(orig)
        assert (n_apply[0] > 0) == fused, n_apply
        res[fused] = tr.flat.flat_param.clone()
        del tr
    assert torch.isfinite(res[True]).all()
    assert torch.equal(res[True], res[False]), "%d parameters differ" % int((res[True] != res[False]).sum())


def test_decoder_weight_gradients_on_the_side_stream_are_bit_identical(fdtune):
    """functional.enable_side_wgrad (default for the depth decoder): its weight gradients, slab reductions and bias sums run on a side
    stream beside the data gradients of the following layers.  Same kernels, same accumulation targets: the gradient buffer after a
    backward pass must equal the all-on-one-stream result bit for bit, also under background GPU load (a tensor freed or reused on
    the main stream while the side stream still reads it would show up here), and the side stream must really have been used."""
    from fusiondepth_amd import functional as FD
    from fusiondepth_amd.trainer import Trainer
    B, H, W = 2, 64, 96
    inp, noise = _batch(B, H, W, 991)
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    grads = {}
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (512, 2048)]
    for mode in ("none", "depth"):
        fdtune.host(side_wgrad=() if mode == "none" else (mode,))
        torch.manual_seed(4321)                                  # the same initial weights for both trainers
        tr = Trainer(_opts(batch_size=B), verbose=False)
        marked = [p for p in tr.models["depth"].parameters() if getattr(p, "_fd_side_wgrad", False)]
        assert (len(marked) > 0) == (mode == "depth")
        for rep in range(3):
            tr.flat.zero_grad()
            saved = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in tr.models.items()}
            torch.cuda.synchronize()
            with torch.cuda.stream(bg):
                for i in range(10 + 9 * rep):
                    junk[i % 2] @ junk[i % 2]
            outputs, losses = tr.process_batch(ginp, groups=tr.accumulate_step)
            losses["loss"].backward()
            if mode == "depth":
                assert len(FD._WGRAD_KEEPALIVE) >= len(marked) // 2          # the side stream has pending work to be joined
            tr._join_side_streams()
            assert not FD._WGRAD_KEEPALIVE
            torch.cuda.synchronize()
            g = tr.flat.flat_grad.clone()
            with torch.no_grad():
                for k, m in tr.models.items():
                    for n, b in m.named_buffers():
                        b.copy_(saved[k][n])
            if mode in grads:
                assert torch.equal(g, grads[mode]), "%s, repetition %d" % (mode, rep)
            grads[mode] = g
    assert float(grads["none"].abs().max()) > 0
    assert torch.equal(grads["none"], grads["depth"]), "%d gradient entries differ" % int((grads["none"] != grads["depth"]).sum())


def test_training_trajectory_is_reproducible_across_runs():
    """Three optimiser steps (forward, backward, Adam, batched weight re-layout, stream forks / joins across step boundaries)
    repeated from the same initial state under background GPU load land on bit-identical parameters."""
    B, H, W = 2, 64, 96
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import functional as FD
    batches = []
    for i in range(3):
        inp, noise = _batch(B, H, W, 1010 + i)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        batches.append(g)
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (512, 1536, 2560)]
    finals = []
    for run in range(3):
        torch.manual_seed(4321)
        tr = Trainer(_opts(batch_size=B), verbose=False)
        for step, b in enumerate(batches):
            with torch.cuda.stream(bg):
                for i in range(10 + 11 * run + 3 * step):
                    junk[(i + run) % 3] @ junk[(i + run) % 3]
            tr.train_step([b])
        torch.cuda.synchronize()
        finals.append(tr.flat.flat_param.clone())
        del tr
    assert torch.equal(finals[0], finals[1]) and torch.equal(finals[0], finals[2])


def test_train_loop_schedule(tmp_path):
    """Trainer(opts).train() over an iterable of reference-schema batches (trainer.py:219-266): optimiser steps per epoch,
    the batch counter, StepLR, the checkpoint cadence, the logging / validation cadence and resume from the written checkpoint."""
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import synthetic
    opt = _opts(height=64, width=96, batch_size=2, log_dir=str(tmp_path), num_epochs=2, save_frequency=1, log_frequency=2)
    tr = Trainer(opt, verbose=False)
    tr.opt.num_epochs = 2                                   # (derived: (8 * 17) // batch_size = 68 epochs, trainer.py:28)
    tr.scheduler_step_size = 1                              # StepLR drops after every epoch
    acc = tr.accumulate_step
    batches = [synthetic.make_batch(tr.batch_size, 64, 96, seed=20 + i) for i in range(3 * acc)]
    for i, b in enumerate(batches):          # KITTI-sized sparse ground truth (the Garg crop is hard-wired to 375x1242)
        gt = torch.rand(tr.batch_size, 1, 375, 1242, generator=torch.Generator().manual_seed(40 + i)) * 60.0 + 2.0
        b["depth_gt"] = (gt * (torch.rand(gt.shape, generator=torch.Generator().manual_seed(80 + i)) < 0.05)).cuda()
    lr0 = tr.lr
    calls = {"val": 0}
    val_orig = tr.val

    def counting_val(loader, save_best=True):
        calls["val"] += 1
        return val_orig(loader, save_best)
    tr.val = counting_val
    tr.train(batches, [batches[0]])
    assert tr.adam_step_count == 2 * 3 and tr.step == 2 * 3 * acc
    assert abs(tr.lr - lr0 * 1e-2) < 1e-12 * lr0 and abs(float(tr.adam_state[1]) - tr.lr) < 1e-9
    models = tmp_path / "mdp" / "models"
    assert sorted(p.name for p in models.iterdir() if p.name.startswith("weights_")) >= ["weights_0", "weights_1"]
    import json
    recs = [json.loads(l) for l in open(tmp_path / "mdp" / "train" / "scalars.jsonl")]
    due = [(e, w) for e in range(2) for w in range(3)
           if any(((w * acc + k) % 2 == 0 and (e * 3 + w) * acc + k < 2000) or ((e * 3 + w) * acc + k) % 2000 == 0 for k in range(acc))]
    assert len(recs) == len(due) == calls["val"] and all(np.isfinite(r["loss"]) and "de/abs_rel" in r for r in recs)
    assert os.path.isfile(tmp_path / "mdp" / "val" / "scalars.jsonl") and os.path.isfile(models / "opt.json")
    # resume: moments, step count and the decayed learning rate come back, and the next step is the one the first run would take
    opt2 = _opts(height=64, width=96, batch_size=2, log_dir=str(tmp_path / "resume"), train_load_weights_folder=str(models / "weights_1"))
    tr2 = Trainer(opt2, verbose=False)
    assert tr2.adam_step_count == tr.adam_step_count and abs(tr2.lr - tr.lr) < 1e-15 and float(tr2.adam_state[0]) == float(tr.adam_step_count)
    assert torch.equal(tr2.flat.flat_param, tr.flat.flat_param) and torch.equal(tr2.exp_avg, tr.exp_avg)
    mb = batches[:acc]
    for b in mb:
        b["_noise"] = [torch.zeros(tr.batch_size, 2, 64, 96, device="cuda") for _ in range(4)]
    tr.set_train(); tr2.set_train()
    la, lb = tr.train_step(mb), tr2.train_step(mb)
    assert_close(float(lb["loss"]), float(la["loss"]), rtol=1e-6, atol=0, what="loss of the step after resume")
    assert_close(cpu(tr2.flat.flat_param), cpu(tr.flat.flat_param), rtol=1e-5, atol=1e-8, what="parameters after the resumed step")


def test_adam_checkpoint_is_the_reference_optimizer_layout(tmp_path):
    """adam.pth must load into torch.optim.Adam over the same parameter list (trainer.py:714-715, 740-744) and back."""
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import synthetic
    tr = Trainer(_opts(height=64, width=96, log_dir=str(tmp_path)), verbose=False)
    mb = [synthetic.make_batch(tr.batch_size, 64, 96, seed=7 + i) for i in range(tr.accumulate_step)]
    tr.train_step(mb)
    st = tr.optimizer_state_dict()
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in tr.parameters_to_train], tr.learning_rate)
    ref_opt.load_state_dict(st)                                              # what the reference's load_model does
    back = ref_opt.state_dict()
    i = max(back["state"])
    assert float(back["state"][i]["step"]) == 1.0 and back["param_groups"][0]["lr"] == tr.lr
    tr.exp_avg.zero_(); tr.adam_step_count = 0; tr.adam_state[0] = 0.0
    tr.load_optimizer_state_dict(back)
    assert tr.adam_step_count == 1 and float(tr.adam_state[0]) == 1.0
    o = tr.flat.offsets[i]
    assert torch.equal(tr.exp_avg[o:o + tr.flat.params[i].numel()].cpu(), back["state"][i]["exp_avg"].reshape(-1).cpu())


@pytest.mark.parametrize("H,W,B,groups", [(128, 192, 2, 1), (192, 640, 6, 2)])
def test_full_step_outputs_against_float64(H, W, B, groups):
    """North-star tolerance at trainer level: ("disp", s), ("depth", 0, s) and every loss of one full training forward, compared
    element-wise (relative) with the oracle's graph evaluated in float64.  The bound per tensor is 1e-4, or twice the error the
    reference's own arithmetic - the float32 oracle - has against float64 on that tensor (train-mode BatchNorm over a handful of
    samples amplifies rounding), whichever is larger; the measured errors are printed.
    Second case = BASELINE.json config 2 exactly: ResNet-18, 640x192, --batch_size 12 = 2 micro-batches of 6, run as ONE stacked
    pass (grouped BatchNorm, per-group SI-log loss) and checked per micro-batch against separate float64 passes."""
    import conftest
    opt = _opts(height=H, width=W, batch_size=B * groups)
    tr, ot = _make_pair(opt)
    assert tr.batch_size == B and tr.accumulate_step == groups
    mbs = [_batch(B, H, W, 700 + g) for g in range(groups)]
    m64 = _float64_models(ot.models)
    ref32, ref64 = [], []
    for inp, noise in mbs:
        with torch.no_grad():
            ref32.append(OT.process_batch(ot.opt, ot.models, {k: v.clone() for k, v in inp.items()}, noise))
            ref64.append(OT.process_batch(ot.opt, m64, _to64(inp), [n.double() for n in noise]))
    ginps = []
    for inp, noise in mbs:
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        ginps.append(g)
    with torch.no_grad():
        if groups == 1:
            outs_g, losses_g = tr.process_batch(ginps[0])
        else:
            outs_g, losses_g = tr.process_batch(tr.stack_micro_batches(ginps), groups=groups)
    worst = 0.0
    for g in range(groups):
        sl = slice(g * B, (g + 1) * B)
        for s in range(4):
            for key in (("disp", s), ("depth", 0, s)):
                r64, r32 = ref64[g][0][key].numpy(), ref32[g][0][key].numpy()
                e_hip, e_ref = _rel_err(cpu(outs_g[key][sl]), r64), _rel_err(r32, r64)
                worst = max(worst, e_hip)
                print("[vs float64] group %d %-16s HIP %.2e | float32 oracle %.2e" % (g, key, e_hip, e_ref))
                conftest.report("R18 %dx%d b%dx%d micro-batch %d %s" % (W, H, B, groups, g, key), e_hip, max(1e-4, 2 * e_ref),
                                "(float32 oracle %.2e)" % e_ref)
                assert e_hip <= max(1e-4, 2 * e_ref), "%s (micro-batch %d): HIP %.3g vs float32 oracle %.3g" % (key, g, e_hip, e_ref)
    # losses: the stacked pass reports sum_g loss_g / groups (trainer.py:237-248 accumulates loss / accumulate_step)
    for k in ref64[0][1]:
        r64 = sum(float(r[1][k]) for r in ref64) / groups
        r32 = sum(float(r[1][k]) for r in ref32) / groups
        e_hip, e_ref = abs(float(losses_g[k]) - r64) / abs(r64), abs(r32 - r64) / abs(r64)
        print("[vs float64] %-16s HIP %.2e | float32 oracle %.2e" % (k, e_hip, e_ref))
        conftest.report("R18 %dx%d b%dx%d %s" % (W, H, B, groups, k), e_hip, max(1e-4, 2 * e_ref), "(float32 oracle %.2e)" % e_ref)
        assert e_hip <= max(1e-4, 2 * e_ref), "%s: HIP %.3g vs float32 oracle %.3g" % (k, e_hip, e_ref)
    print("worst element-wise relative error of disp / depth vs float64: %.2e" % worst)


def test_twenty_step_trajectory_vs_oracle_fixture(golden):
    """20 optimiser steps from the same initial state and the same seeded batches as
    tests/golden/make_trajectory.py (float32 CPU oracle).  Step 0 must agree to 2e-4; later steps carry Adam's amplification of
    rounding differences (its first updates are ~lr * sign(g)), which the fixture quantifies itself: it also holds the float64
    trajectory of the same graph (the two oracle runs are up to 1.1e-2 apart within these 20 steps): the HIP run must stay within
    2.5x the largest float32-vs-float64 gap of the oracle (two builds of this package that differ only in the summation order of
    the small-plane BatchNorm reach 1.0e-2 and 1.6e-2 at step 16), with step 0 at 2e-4 and step 1 at 1e-3."""
    g = golden("trajectory_r18_128x192_b2")
    opt = _opts(height=128, width=192)
    tr, _ = _make_pair(opt)
    B, H, W = 2, 128, 192
    traj = {}
    for step in range(20):
        inp, noise = _batch(B, H, W, 900 + step)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        losses = tr.train_step([ginp])
        for k, v in losses.items():
            traj.setdefault(k.replace("/", "_"), []).append(float(v))
    assert set(traj) == {k[4:] for k in g if k.startswith("f32/") and not k.startswith("f32/param")}
    f32, f64, got = g["f32/loss"], g["f64/loss"], np.asarray(traj["loss"])
    drift = np.abs(f32 - f64) / f64
    err = np.abs(got - f64) / f64
    print("loss, HIP vs float64 fixture:      " + " ".join("%.1e" % e for e in err))
    print("loss, float32 vs float64 fixture:  " + " ".join("%.1e" % e for e in drift))
    assert np.isfinite(got).all()
    assert err[0] <= 2e-4 and err[1] <= 1e-3
    assert (err <= 2.5 * drift.max()).all(), "trajectory leaves the band of the reference's own float32-vs-float64 drift"
    for k in traj:
        assert_close(np.asarray(traj[k])[:1], g["f32/" + k][:1], rtol=2e-4, atol=1e-6, what="step 0 " + k)
    assert tr.adam_step_count == 20


def test_weight_layout_cache_forgets_dead_trainers():
    """the process-global layout cache must not keep the layouts (and re-layout jobs) of a deleted Trainer alive
    for ever, and must not free buffers a captured graph may still read: entries of dead parameters move to a retired list."""
    import gc
    from fusiondepth_amd import functional as FD, synthetic
    from fusiondepth_amd.trainer import Trainer
    FD.evict_dead_weight_layouts(); FD.release_retired_layouts()
    tr = Trainer(_opts(), verbose=False)
    mb = [synthetic.make_batch(tr.batch_size, 64, 96, seed=3 + i) for i in range(tr.accumulate_step)]
    tr.train_step(mb); tr.train_step(mb)
    ids = {p._fd_cache_id for p in tr.parameters_to_train if hasattr(p, "_fd_cache_id")}
    mine = [k for k in FD._WT_CACHE if k[0] in ids]
    assert len(mine) > 50 and FD._WT_PLAN[0] is not None
    del tr, mb
    gc.collect()
    tr2 = Trainer(_opts(), verbose=False)                 # evicts on construction
    assert not [k for k in FD._WT_CACHE if k[0] in ids], "layouts of the deleted trainer are still cached"
    assert FD._WT_PLAN[0] is None and len(FD._WT_RETIRED) >= len(mine)
    mb = [synthetic.make_batch(tr2.batch_size, 64, 96, seed=3 + i) for i in range(tr2.accumulate_step)]
    losses = tr2.train_step(mb)
    assert np.isfinite(float(losses["loss"]))
    assert FD.release_retired_layouts() > 0
