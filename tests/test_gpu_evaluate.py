"""Device-side evaluate_depth core (fusiondepth_amd/evaluate_depth.py) vs the reference goldens and the numpy oracle."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_compute_errors_vs_reference_golden(golden):
    from fusiondepth_amd import evaluate_depth as ED
    g = golden("evaluate_metrics")
    for i, (gt, pred) in enumerate(gin.eval_pairs(int(g["seed"]))):
        got = ED.compute_errors(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda())
        assert_close(np.array(got), g["errors%d" % i], rtol=2e-5, atol=0, what="compute_errors %d" % i)


def test_post_process_disparity_bit_exact_vs_reference_golden(golden):
    from fusiondepth_amd import evaluate_depth as ED
    from oracle import evaluate as OE
    g = golden("evaluate_metrics")
    l, r = gin.disp_pair(4243, 2, 192, 640)
    pp = ED.batch_post_process_disparity(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())
    assert pp.dtype == torch.float64
    pp = pp.cpu().numpy()
    assert np.array_equal(pp[:, ::7, ::3], g["pp_sub"])
    assert np.array_equal(np.concatenate([pp[:, :, :40], pp[:, :, -40:]], 2)[:, ::16], g["pp_edges"])
    assert np.array_equal(pp, OE.batch_post_process_disparity(l, r))
    l2, r2 = gin.disp_pair(5, 3, 33, 41)                       # odd sizes
    got = ED.batch_post_process_disparity(torch.from_numpy(l2).cuda(), torch.from_numpy(r2).cuda()).cpu().numpy()
    assert np.array_equal(got, OE.batch_post_process_disparity(l2, r2))


@pytest.mark.parametrize("split,median,factor", [("eigen", True, 1.0), ("eigen", False, 5.4), ("eigen_benchmark", True, 1.0)])
def test_evaluate_predictions_vs_oracle(split, median, factor):
    """The per-image evaluation loop (evaluate_depth.py:344-478) on KITTI-sized ground truth of two different drive sizes."""
    from fusiondepth_amd import evaluate_depth as ED
    from oracle import evaluate as OE
    rng = np.random.RandomState(77)
    gts, disps = [], []
    for (gh, gw) in ((375, 1242), (370, 1226), (375, 1242)):
        gt = rng.uniform(1.5, 90.0, (gh, gw)).astype(np.float32)
        gt[rng.rand(gh, gw) > 0.05] = 0.0
        gts.append(gt)
        disps.append(rng.uniform(0.02, 0.6, (192, 640)).astype(np.float32))
    want, want_r = OE.evaluate_predictions(disps, gts, split, factor, not median)
    got, got_r = ED.evaluate_predictions(torch.from_numpy(np.stack(disps)).cuda(), gts, split, factor, not median)
    assert_close(got[:4], want[:4], rtol=5e-5, atol=0, what="mean abs_rel / sq_rel / rmse / rmse_log")
    # a1-a3 are fractions of ~10^4 pixels under a threshold: a pixel within rounding of 1.25^k may fall on either side
    assert_close(got[4:], want[4:], rtol=0, atol=3e-4, what="mean a1 / a2 / a3")
    assert_close(got_r, want_r, rtol=1e-5, atol=0, what="ratios")
    assert (len(got_r) == 3) == median


@pytest.mark.parametrize("hin,win,hout,wout", [(192, 640, 375, 1242), (192, 640, 370, 1226), (320, 1024, 375, 1242), (33, 41, 17, 90), (8, 8, 8, 8)])
def test_resize_linear_cv_rule_is_the_oracle_restatement(hin, win, hout, wout):
    """fd_resize_linear_cv == oracle.evaluate.resize_bilinear bit for bit: both restate OpenCV's float32 INTER_LINEAR (coefficient rule
    in double -> float, horizontal pass, vertical pass) - up- and down-sampling, identity.  (Unpinned against OpenCV itself: not
    installed in the build image.)"""
    from fusiondepth_amd import functional as FD
    from oracle import evaluate as OE
    img = np.random.RandomState(hin + wout).uniform(0.01, 0.7, (2, hin, win)).astype(np.float32)
    got = FD.resize_linear_cv(torch.from_numpy(img).cuda(), (hout, wout)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], OE.resize_bilinear(img[i], hout, wout)), np.abs(got[i] - OE.resize_bilinear(img[i], hout, wout)).max()
