"""GPU parity of the HIP-backed Refiner (BASELINE.json config 5 / SURVEY.md §8f rank 1) against the golden produced by the
reference's own Refiner.process_batch / compute_losses, and against the CPU oracle over optimiser steps."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_close, check_grad_compact
from oracle import refiner as OR

pytestmark = pytest.mark.gpu


def _make(B=1, H=192, W=640):
    """The frozen stage-1 networks reach the Refiner the way the reference feeds them (refiner.py:56-60, 84-152): as a
    ``Trainer.save_model`` folder (encoder.pth with the height / width / use_stereo extras) given by --refine_load_weights_folder;
    ``refine2d_decoder.pth`` is there too (the resume case)."""
    import tempfile
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.refiner import Refiner
    oopt = OR.default_opt(batch_size=B, height=H, width=W)
    omodels = gin.refiner_models(OR.build_models(oopt, 0))
    folder = tempfile.mkdtemp(prefix="fd_stage1_")
    for k, m in omodels.items():
        sd = {n: v.detach().clone() for n, v in m.state_dict().items()}
        if k == "encoder":
            sd.update(height=H, width=W, use_stereo=False)
        torch.save(sd, "%s/%s.pth" % (folder, k))
    o = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", str(B),
                                  "--height", str(H), "--width", str(W), "--refine_load_weights_folder", folder])
    rf = Refiner(o, verbose=False)
    oopt = OR.default_opt(batch_size=B, height=H, width=W, learning_rate=o.learning_rate)
    for k, m in omodels.items():
        for name, t in rf.models[k].state_dict().items():
            assert torch.equal(t.cpu(), m.state_dict()[name]), "refine_load_weights_folder: %s.%s was not loaded" % (k, name)
    return rf, oopt, omodels


def test_refiner_needs_its_stage1_folder(tmp_path):
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.refiner import Refiner
    o = MonodepthOptions().parse(["--num_layers", "18", "--height", "64", "--width", "96", "--refine_load_weights_folder",
                                  str(tmp_path / "missing")])
    with pytest.raises(AssertionError, match="Cannot find a folder"):
        Refiner(o, verbose=False)
    (tmp_path / "empty").mkdir()
    o.refine_load_weights_folder = str(tmp_path / "empty")
    with pytest.raises(FileNotFoundError):
        Refiner(o, verbose=False)


def test_refiner_step_vs_reference_golden(golden):
    g = golden("refiner_b1_192x640")
    B, H, W = 1, 192, 640
    rf, oopt, _ = _make(B, H, W)
    inp, _ = gin.refiner_inputs(808, B, H, W)
    torch.manual_seed(int(g["noise_seed"]))
    noise = [torch.randn(B, 2, H, W) for _ in range(4)]
    np.testing.assert_array_equal(noise[0].numpy().reshape(-1)[:16], g["noise_head"])
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    outputs, losses = rf.process_batch(ginp)
    for k, v in losses.items():
        assert_close(float(v.detach()), float(g["L/" + k.replace("/", "_")]), rtol=2e-4, atol=1e-7, what=k)
    for s in range(4):
        assert_close(outputs[("disp", s)].detach().cpu().numpy(), g["disp%d" % s], rtol=1e-3, atol=1e-4, what="refined disp%d" % s)
    losses["loss"].backward()
    for n, p in rf.models["refine2d_decoder"].named_parameters():
        if ("g/" + n) in g or ("g/" + n + "@sum") in g:
            check_grad_compact(g, "g/" + n, p.grad.cpu().numpy(), rtol=2e-3, atol=1e-5)


def test_refiner_trains_like_the_oracle():
    """Three Adam steps of the refine decoder: loss trajectory vs the oracle harness (same init, inputs, noise)."""
    B, H, W = 1, 192, 640
    rf, oopt, omodels = _make(B, H, W)
    opt_o = torch.optim.Adam(omodels["refine2d_decoder"].parameters(), rf.lr)
    traj = []
    for step in range(3):
        inp, noise = gin.refiner_inputs(820 + step, B, H, W)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        _, lo = OR.process_batch(oopt, omodels, inp, noise)
        opt_o.zero_grad()
        lo["loss"].backward()
        opt_o.step()
        lg = rf.train_step(ginp)
        traj.append((float(lg["loss"]), float(lo["loss"])))
    print("refiner loss (HIP, oracle):", traj)
    assert np.isfinite(traj).all()
    assert_close(traj[0][0], traj[0][1], rtol=2e-4, atol=0, what="refiner loss at step 0")
    assert_close([t[0] for t in traj], [t[1] for t in traj], rtol=5e-3, atol=0, what="refiner loss trajectory")


def test_refiner_steps_are_reproducible_under_gpu_contention():
    """Two optimiser steps of the refiner from the same state, repeated with different background GPU load: bit-identical
    refine-decoder parameters (the refiner drives frozen encoders on side streams, a median-scaling pass and the fused loss)."""
    B, H, W = 1, 192, 640
    batches = []
    for step in range(2):
        inp, noise = gin.refiner_inputs(840 + step, B, H, W)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        batches.append(g)
    bg = torch.cuda.Stream()
    junk = [torch.randn(s, s, device="cuda") for s in (768, 2048)]
    finals = []
    for run in range(3):
        rf, _, _ = _make(B, H, W)
        for step, b in enumerate(batches):
            with torch.cuda.stream(bg):
                for i in range(8 + 9 * run + 2 * step):
                    junk[i % 2] @ junk[i % 2]
            rf.train_step({k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in b.items()})
        torch.cuda.synchronize()
        finals.append(torch.cat([p.detach().reshape(-1) for p in rf.models["refine2d_decoder"].parameters()]).clone())
        del rf
    assert torch.equal(finals[0], finals[1]) and torch.equal(finals[0], finals[2])


def test_refiner_prefetched_frozen_block_changes_nothing(fdtune):
    """``train_step(inputs, next_inputs)`` issues the next batch's frozen forward passes on their own stream beside this step's
    refine-decoder work (Refiner.prefetch_frozen).  Six steps over three batches, with background load on another stream: losses and
    refine-decoder parameters bit-identical to the plain ``train_step(inputs)`` loop; a caller that passes a DIFFERENT batch than it
    announced gets that batch's own (recomputed) result."""
    B, H, W = 2, 192, 640
    batches = []
    for k in range(3):
        inp, noise = gin.refiner_inputs(860 + k, B, H, W)
        g = {k_: v.cuda() for k_, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        batches.append(g)
    order = [0, 1, 2, 0, 2, 1]
    bg = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device="cuda")

    def run(prefetch, lie=False):
        fdtune.host(refiner_prefetch=prefetch)
        rf, _, _ = _make(B, H, W)
        losses = []
        for i, b in enumerate(order):
            with torch.cuda.stream(bg):
                for _ in range(6 + 5 * i):
                    junk @ junk
            nxt = batches[order[i + 1]] if i + 1 < len(order) else None
            if lie and i == 2:
                nxt = batches[order[i]]                 # announced batch != the batch of the next call
            losses.append(rf.train_step(batches[b], nxt)["loss"].detach().clone())
        torch.cuda.synchronize()
        return torch.stack(losses), torch.cat([p.detach().reshape(-1) for p in rf.models["refine2d_decoder"].parameters()]).clone()

    l0, p0 = run(False)
    l1, p1 = run(True)
    l2, p2 = run(True, lie=True)
    assert torch.equal(l0, l1) and torch.equal(p0, p1), "prefetched frozen block changes the step"
    assert torch.equal(l0, l2) and torch.equal(p0, p2), "a mis-announced next batch must be recomputed"


@pytest.mark.parametrize("layers", [18, 50])
def test_frozen_batchnorm_folded_into_the_convolution(layers, fdtune):
    """functional.conv_bn_frozen: the eval-mode BatchNorms of a frozen ResNet that have no residual input (bn1 of every block, the
    downsample branches; bn1 / bn2 of a Bottleneck) are folded into their convolutions.  Features vs the two-launch form: relative L2
    <= 2e-6 per level (rounding of w * a only); new statistics (load_state_dict) are picked up."""
    from fusiondepth_amd import networks
    import fusiondepth_amd.functional as FD
    torch.manual_seed(layers)
    enc = networks.ResnetEncoder(layers, False).cuda().eval()
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    for p in enc.parameters():
        p.requires_grad_(False)
    FD.enable_weight_cache(list(enc.parameters()), frozen=True)
    x = torch.rand(2, 3, 64, 96, device="cuda")

    def both():
        with torch.no_grad():
            fdtune.host(fold_frozen_bn=False)
            plain = [f.clone() for f in enc(x)]
            fdtune.host(fold_frozen_bn=True)
            folded = [f.clone() for f in enc(x)]
        return plain, folded
    plain, folded = both()
    n0 = len(FD._FOLDED)
    assert n0 >= (11 if layers == 18 else 36)
    for i, (a, b) in enumerate(zip(plain, folded)):
        err = float((a - b).norm() / a.norm())
        assert err <= 2e-6, "features[%d]: folded vs two-launch BatchNorm relative L2 %.3e" % (i, err)
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.mul_(0.5)                       # version counter bumps: the folded pairs are stale
    plain2, folded2 = both()
    assert float((plain2[-1] - plain[-1]).norm() / plain[-1].norm()) > 1e-3
    for a, b in zip(plain2, folded2):
        assert float((a - b).norm() / a.norm()) <= 2e-6


def test_refiner_train_loop_logs_depth_metrics(tmp_path):
    """Refiner.train() / run_epoch (refiner.py:264-297) over batches that carry ``depth_gt``: the logged batches go through
    compute_depth_losses, which reads ("depth", 0, 0) - derived lazily from the refined disparity (the Refiner's
    ``Outputs`` had no depth_spec and the first logged batch raised KeyError)."""
    import json
    B, H, W = 1, 192, 640
    rf, _, _ = _make(B, H, W)
    rf.opt.log_dir, rf.log_path = str(tmp_path), str(tmp_path / "rf")
    rf.opt.num_epochs, rf.opt.log_frequency, rf.opt.save_frequency = 1, 1, 1
    batches = []
    for i in range(2):
        inp, noise = gin.refiner_inputs(860 + i, B, H, W)
        g = {k: v.cuda() for k, v in inp.items()}
        g["_noise"] = [n.cuda() for n in noise]
        gt = torch.rand(B, 1, 375, 1242, generator=torch.Generator().manual_seed(60 + i)) * 50.0 + 3.0
        g["depth_gt"] = gt.cuda()
        batches.append(g)
    rf.train(batches)
    recs = [json.loads(l) for l in open(tmp_path / "rf" / "train" / "scalars.jsonl")]
    assert len(recs) == 2 and all(np.isfinite(r["loss"]) and np.isfinite(r["de/abs_rel"]) and r["de/abs_rel"] > 0 for r in recs)
    assert rf.adam_step_count == 2


# ---- refine-decoder input construction on libfdhip (csrc/refine.hip) --------------------------------------
def _refine_case(seed, B, H, W, dense=False):
    """Disparity pyramid, sparse LiDAR map + its 2-channel scatter, intrinsics per scale - on the CPU."""
    rng = np.random.RandomState(seed)
    disps = [torch.from_numpy(rng.uniform(0.01, 0.6, (B, 1, H >> s, W >> s)).astype(np.float32)) for s in range(4)]
    beam = torch.zeros(B, 1, H, W)
    if dense:        # an r200-like map: returns everywhere, also outside the crop, ties between values
        m = torch.from_numpy(rng.rand(B, 1, H, W) < 0.3)
        beam[m] = torch.from_numpy(np.round(rng.uniform(0.03, 0.8, int(m.sum())), 2).astype(np.float32))
    else:
        rows = [int(H * f) for f in (0.45, 0.55, 0.7, 0.85)]
        for r in rows:
            beam[:, 0, r, 3:W - 3:3] = torch.from_numpy(rng.uniform(0.05, 0.65, (B, len(range(3, W - 3, 3)))).astype(np.float32))
    two = torch.from_numpy(rng.rand(B, 2, H, W).astype(np.float32)) * (torch.from_numpy(rng.rand(B, 2, H, W)) < 0.2)
    inv_K = {}
    for s in range(4):
        K = np.array([[0.58 * (W >> s), 0, 0.5 * (W >> s), 0], [0, 1.92 * (H >> s), 0.5 * (H >> s), 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        inv_K[s] = torch.from_numpy(np.linalg.pinv(K)).unsqueeze(0).repeat(B, 1, 1)
    return disps, beam, two.float(), inv_K


@pytest.mark.parametrize("dense", [False, True])
def test_masked_median_is_torch_median(dense):
    """fd_masked_median (radix select) == torch.median(x[mask] * scale): the lower median over the whole batch, bit for bit - even /
    odd counts, ties, negative values, one element, an empty selection (NaN), a NaN in the selection (NaN)."""
    from fusiondepth_amd import functional as FD
    rng = np.random.RandomState(3 + dense)
    for B, H, W, win in ((2, 192, 640, (78, 190, 23, 617)), (3, 40, 70, None), (1, 17, 33, (2, 15, 1, 30))):
        x = torch.from_numpy(rng.randn(B, 1, H, W).astype(np.float32))
        if dense:
            x = torch.round(x * 4) / 4               # many equal values
        gate = torch.from_numpy((rng.rand(B, 1, H, W) < (0.4 if dense else 0.01)).astype(np.float32))
        for drop in (0, 1):                           # an even and an odd selection size
            y0, y1, x0, x1 = win if win is not None else (0, H, 0, W)
            crop = torch.zeros_like(gate, dtype=torch.bool)
            crop[:, :, y0:y1, x0:x1] = True
            m = (gate > 0) & crop
            if drop:
                first = m.flatten().nonzero()[0]
                gate.view(-1)[first] = 0
                m = (gate > 0) & crop
            want = torch.median(x[m] * 2.5)
            got = FD.masked_median(x.cuda(), gate.cuda(), 2.5, win)
            assert got.cpu().item() == want.item(), (B, H, W, int(m.sum()), got.item(), want.item())
    x = torch.tensor([[[[3.0, -1.0, 7.0, 2.0]]]])
    g = torch.tensor([[[[0.0, 1.0, 0.0, 0.0]]]])
    assert FD.masked_median(x.cuda(), g.cuda()).item() == -1.0
    assert np.isnan(FD.masked_median(x.cuda(), torch.zeros_like(g).cuda()).item())
    xn = x.clone(); xn[0, 0, 0, 1] = float("nan")
    assert np.isnan(FD.masked_median(xn.cuda(), torch.ones_like(g).cuda()).item())


@pytest.mark.parametrize("B,H,W,dense", [(2, 192, 640, False), (1, 192, 640, True), (2, 320, 1024, False)])
@pytest.mark.parametrize("catxy,a0", [("true", "true"), ("true", "false"), ("false", "true")])
def test_refine_inputs_vs_oracle(B, H, W, dense, catxy, a0):
    """fd_refine_inputs (four launches) against the oracle's restatement of refiner.py:316-348 (ATen on the CPU): the median ratio
    of every scale exactly up to the rounding of the depths it selects from, every output channel to 1e-5 relative (pooled channels
    exact)."""
    from fusiondepth_amd import functional as FD
    disps, beam, two, inv_K = _refine_case(11 + B + H, B, H, W, dense)
    opt = OR.default_opt(batch_size=B, height=H, width=W, catxy=catxy, refine_a0=a0)
    inputs = {"4beam": beam, "2channel": two}
    inputs.update({("inv_K", s): inv_K[s] for s in range(4)})
    want = OR.refine_inputs(opt, inputs, {("disp", s): disps[s] for s in range(4)})
    got, stats = FD.refine_inputs([d.cuda() for d in disps], beam.cuda(), two.cuda(), [inv_K[s].cuda() for s in range(4)], H, W,
                                  opt.min_depth, opt.max_depth, catxy=(catxy == "true"), pool_disp0=(a0 == "true"), return_stats=True)
    stats = stats.cpu().numpy()
    mask = beam > 0
    crop = torch.zeros_like(mask); crop[:, :, 78:190, 23:617] = 1
    mask = mask * crop
    assert stats[0, 3] == int(mask.sum()) and stats[0, 2] == torch.median(beam[mask] * 100.0).item()
    for s in range(4):
        w, g = want[("disp", s)].numpy(), got[s].cpu().numpy()
        assert w.shape == g.shape
        nch = w.shape[1]
        assert np.array_equal(g[:, nch - 2:], w[:, nch - 2:]), "pooled 2-channel map at scale %d" % s
        assert_close(g[:, 0], w[:, 0], rtol=2e-5, atol=2e-6, what="scaled disparity at scale %d" % s)
        if catxy == "true":
            assert_close(g[:, 1:4], w[:, 1:4], rtol=2e-5, atol=2e-5, what="Cat_xy channels at scale %d" % s)


def test_refiner_inputs_path_has_no_aten_kernels():
    """Between the frozen networks and the refine decoder nothing but libfdhip launches: Refiner.refine_inputs equals its ATen form
    of rounds 2-4 (refine_inputs_aten) on a real batch."""
    rf, oopt, _ = _make(B=2)
    from fusiondepth_amd import synthetic
    inp = synthetic.make_batch(2, 192, 640, seed=5)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    with torch.no_grad():
        feats = rf.models["encoder"](inp["color_aug", 0, 0])
        outputs = rf.models["depth"](feats, beam_features=rf.models["beam_encoder"](inp["2channel"])) if rf.opt.refine_depthnet_with_beam == "true" \
            else rf.models["depth"](feats)
        a = rf.refine_inputs(inp, outputs)
        b = rf.refine_inputs_aten(inp, outputs)
    for s in rf.opt.scales:
        assert_close(a[("disp", s)].cpu().numpy(), b[("disp", s)].cpu().numpy(), rtol=2e-5, atol=2e-5, what="refine inputs at scale %d" % s)


def test_channel_padded_refine_decoder_equals_the_unpadded_one(fdtune):
    """tuning.host.pad_odd_channels: the refine decoder's 262 / 134 / 102 / 22-channel blocks run zero-padded to a multiple of 16 (MFMA
    fast path / Winograd kernels instead of the generic gather GEMM).  Outputs and every parameter gradient must agree with the
    unpadded run to float32 rounding (different kernels, different summation order), and the parameters keep the reference's shapes."""
    from fusiondepth_amd import networks
    rng = np.random.RandomState(41)
    B, H, W = 2, 96, 160
    dec = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4), road=True, catxy=True, deep=True).cuda()
    assert dec._cin_pad == {("upconv", 3, 1): 272, ("upconv", 2, 1): 144, ("upconv", 1, 1): 112, ("upconv", 0, 1): 32}
    assert dec.convs[("upconv", 3, 1)][0].conv.conv.weight.shape == (262, 262, 3, 3)
    feats = [torch.from_numpy(rng.randn(B, c, H >> (s + 1), W >> (s + 1)).astype(np.float32)).cuda() for s, c in enumerate((64, 64, 128, 256, 512))]
    maps = {("disp", s): torch.from_numpy(rng.rand(B, 6, H >> s, W >> s).astype(np.float32)).cuda() for s in range(4)}
    cot = {s: torch.from_numpy(rng.randn(B, 1, H >> s, W >> s).astype(np.float32)).cuda() for s in range(4)}
    res = {}
    for on in (True, False):
        fdtune.host(pad_odd_channels=on)
        dec.zero_grad()
        out = dec(feats, depth_maps=maps, tanh=False)
        sum((out[("disp", s)] * cot[s]).sum() for s in range(4)).backward()
        res[on] = ({s: out[("disp", s)].detach().cpu().numpy() for s in range(4)},
                   {n: p.grad.detach().cpu().numpy().copy() for n, p in dec.named_parameters()})
    for s in range(4):
        assert_close(res[True][0][s], res[False][0][s], rtol=2e-5, atol=2e-6, what="disp %d" % s)
    for n in res[True][1]:
        a, b = res[True][1][n], res[False][1][n]
        assert a.shape == b.shape
        assert np.abs(a - b).sum() <= 2e-5 * np.abs(b).sum() + 1e-9, "gradient of %s: %g of %g" % (n, np.abs(a - b).sum(), np.abs(b).sum())


def _sparse_refiner_inputs(seed, B, H, W, n_points):
    """refiner inputs whose LiDAR maps are r100 / r200 samples (~3.5 - 7 m returns, like gin.refiner_inputs) instead of 4 scan lines"""
    from oracle import scatter as OS
    inp, noise = gin.refiner_inputs(seed, B, H, W)
    roi = (76, 190, 2, 638)
    for i, f in enumerate((0, -1, 1)):
        beam = gin.lidar_random(np.random.RandomState(seed + 30 + i), B, H, W, n_points, roi, lo=3.5, hi=7.0)
        two = np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0], roi)) for b in range(B)])
        inp[("2channel", f, 0)] = torch.from_numpy(two)
        if f == 0:
            inp["2channel"] = torch.from_numpy(two)
            inp["4beam"] = torch.from_numpy(beam)
    return inp, noise


@pytest.mark.parametrize("n_points", [100, 200], ids=["r100", "r200"])
def test_refiner_step_on_random_sample_lidar_vs_oracle(n_points):
    """BASELINE config 5's sparse inputs through a Refiner step ("missing" 4): r100 / r200 maps - 100 / 200 returns per
    image - make the masked medians of refiner.py:316-348 and the SI-log reductions of refiner.py:557-563 run on a few dozen selected
    pixels (the small-count path of the radix select and of the masked sums).  One process_batch + backward against the oracle's
    refiner: every loss, the refined disparities, the gradient of every refine-decoder parameter."""
    import conftest
    B, H, W = 1, 192, 640
    rf, oopt, omodels = _make(B, H, W)
    inp, noise = _sparse_refiner_inputs(860 + n_points, B, H, W, n_points)
    assert int((inp["4beam"] > 0).sum()) == B * n_points
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    outs_o, lo = OR.process_batch(oopt, omodels, inp, noise)
    for p in omodels["refine2d_decoder"].parameters():
        p.grad = None
    lo["loss"].backward()
    outputs, losses = rf.process_batch(ginp)
    assert set(losses) == set(lo)
    for k in lo:
        a, b = float(losses[k].detach()), float(lo[k])
        assert np.isnan(a) == np.isnan(b), "%s: HIP %r, oracle %r" % (k, a, b)
        if not np.isnan(b):
            conftest.report("refiner %s-point LiDAR: %s |HIP - oracle| / |oracle|" % (n_points, k), abs(a - b) / max(abs(b), 1e-30), 3e-4)
            assert_close(a, b, rtol=3e-4, atol=1e-7, what=k)
    for s in range(4):
        assert_close(outputs[("disp", s)].detach().cpu().numpy(), outs_o[("disp", s)].detach().numpy(), rtol=1e-3, atol=1e-4,
                     what="refined disp%d" % s)
    losses["loss"].backward()
    num = den = 0.0
    for (n, p), po in zip(rf.models["refine2d_decoder"].named_parameters(), omodels["refine2d_decoder"].parameters()):
        assert (p.grad is None) == (po.grad is None), n
        if po.grad is None:
            continue
        d = p.grad.cpu().double() - po.grad.double()
        num += float((d * d).sum()); den += float((po.grad.double() ** 2).sum())
    e = (num / den) ** 0.5
    conftest.report("refiner %s-point LiDAR: refine-decoder gradient, relative L2 vs the float32 oracle" % n_points, e, 5e-3)
    assert den > 0 and e <= 5e-3


@pytest.mark.parametrize("H,W", [(96, 320), (128, 416)])
def test_refine_inputs_clamp_the_crop_like_the_reference(H, W):
    """refiner.py:329 ``crop_mask[:, :, 78:190, 23:617] = 1`` is a slice: on a plane smaller than the window it is clamped silently
    (the library rejected the unclamped window).  FD.refine_inputs == the oracle's refine_inputs at 96x320 / 128x416."""
    from fusiondepth_amd import functional as FD
    from oracle import layers as OL
    B = 2
    rng = np.random.RandomState(5)
    disps = [torch.from_numpy(rng.uniform(0.05, 0.9, size=(B, 1, H >> s, W >> s)).astype(np.float32)) for s in range(4)]
    beam = torch.from_numpy(gin.lidar_random(rng, B, H, W, 150, roi=(H // 3, H - 2, 2, W - 2), lo=3.5, hi=7.0))
    two = torch.from_numpy(rng.uniform(0, 1, size=(B, 2, H, W)).astype(np.float32))
    inv_K = [gin.intrinsics(B, H, W, s)[1] for s in range(4)]
    oopt = OR.default_opt(batch_size=B, height=H, width=W, catxy="true", refine_a0="true")
    inputs = {"4beam": beam, "2channel": two}
    inputs.update({("inv_K", s): inv_K[s] for s in range(4)})
    want = OR.refine_inputs(oopt, inputs, {("disp", s): disps[s] for s in range(4)})
    got = FD.refine_inputs([d.cuda() for d in disps], beam.cuda(), two.cuda(), [k.cuda() for k in inv_K], H, W, oopt.min_depth, oopt.max_depth,
                           catxy=True, pool_disp0=True)
    for s in range(4):
        assert_close(got[s].cpu().numpy(), want[("disp", s)].numpy(), rtol=2e-5, atol=2e-5, what="refine inputs scale %d at %dx%d" % (s, H, W))
