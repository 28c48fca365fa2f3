"""HIP LiDAR rasterisation (fd_velo_rasterize) vs the reference golden and the oracle: bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import inputs as gin  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(velo, P, im_h=375, im_w=1242, shape=(384, 1280), vel_depth=False):
    """-> (4-beam map for ``shape``, unpadded full-resolution float64 depth image)."""
    import fusiondepth_amd.functional as FD
    pts = torch.from_numpy(velo).cuda()
    beam = FD.velo_rasterize(pts, P, im_h, im_w, shape, vel_depth=vel_depth)
    full = FD.velo_rasterize(pts, P, im_h, im_w, None, return_full=True, beam=False, vel_depth=vel_depth)
    torch.cuda.synchronize()
    return beam.cpu().numpy(), full.cpu().numpy()


def _write_kitti_files(d, velo):
    """The scan and its calibration in the reference's on-disk formats (kitti_utils.py:8-30, 43-57)."""
    cal = gin.lidar_scan.calib
    fmt = lambda a: " ".join("%.17g" % v for v in np.asarray(a, dtype=np.float64).reshape(-1))
    with open(os.path.join(d, "calib_cam_to_cam.txt"), "w") as f:
        f.write("calib_time: 09-Jan-2012 13:57:47\nS_rect_02: %s\nR_rect_00: %s\nP_rect_02: %s\n"
                % (fmt(cal["S_rect_02"]), fmt(cal["R_rect_00"]), fmt(cal["P_rect_02"])))
    with open(os.path.join(d, "calib_velo_to_cam.txt"), "w") as f:
        f.write("R: %s\nT: %s\n" % (fmt(cal["R"]), fmt(cal["T"])))
    velo.tofile(os.path.join(d, "scan.bin"))
    return os.path.join(d, "scan.bin")


def test_rasterize_matches_reference_golden(golden):
    g = golden("rasterize_scan3")
    velo, P = gin.lidar_scan(int(g["seed"]))
    beam, full = _run(velo, P)
    want = np.zeros((375, 1242))
    want[g["full_rows"], g["full_cols"]] = g["full_vals"]
    assert np.array_equal(full, want), "max |diff| %g at %d pixels" % (np.abs(full - want).max(), (full != want).sum())
    assert beam.dtype == np.float32 and np.array_equal(beam, g["beam"])


@pytest.mark.parametrize("seed,n,shape", [(11, 6000, (384, 1280)), (12, 120000, (384, 1280)), (13, 3000, (352, 1280)), (14, 900, (384, 1281))])
def test_rasterize_matches_oracle(seed, n, shape):
    """Other scans (dense ones pile dozens of points on a pixel), the crop branch and an odd target width (ceil-mode pooling)."""
    from oracle import rasterize as OR
    velo, P = gin.lidar_scan(seed, n_points=n)
    beam, full = _run(velo, P, shape=shape)
    assert np.array_equal(full, OR.depth_image(velo, P, 375, 1242))
    assert np.array_equal(beam, OR.four_beam(velo, P, 375, 1242, shape))
    _, full_vd = _run(velo, P, shape=shape, vel_depth=True)
    assert np.array_equal(full_vd, OR.depth_image(velo, P, 375, 1242, vel_depth=True))


def test_rasterize_is_deterministic_and_order_rule_holds():
    """Atomics make the kernel order independent: repeated runs agree bit for bit; permuting the points changes the answer only
    the way the reference's does (last-write-wins / first-of-group), i.e. it keeps matching the oracle."""
    from oracle import rasterize as OR
    velo, P = gin.lidar_scan(21, n_points=40000)
    a = _run(velo, P)
    b = _run(velo, P)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    perm = np.random.RandomState(0).permutation(velo.shape[0])
    beam, full = _run(velo[perm], P)
    assert np.array_equal(full, OR.depth_image(velo[perm], P, 375, 1242))


def test_rasterize_empty_and_behind():
    _, P = gin.lidar_scan(3)
    beam, full = _run(np.zeros((0, 4), np.float32), P)
    assert not full.any() and not beam.any() and beam.shape == (192, 640)
    beam, full = _run(np.array([[-5.0, 0.0, 0.0, 0.0], [-1.0, 2.0, 0.5, 0.0]], np.float32), P)
    assert not full.any() and not beam.any()


def test_rasterize_feeds_the_scatter(golden):
    """rasterise -> scatter chained on the device == the reference 4-beam map through the oracle scatter."""
    import fusiondepth_amd.functional as FD
    from oracle import scatter as OS
    g = golden("rasterize_scan3")
    velo, P = gin.lidar_scan(int(g["seed"]))
    beam = FD.velo_rasterize(torch.from_numpy(velo).cuda(), P, 375, 1242)
    two = FD.scatter_2channel(beam).cpu().numpy()
    depth, conf = OS.scatter_2channel_c(g["beam"])
    assert np.array_equal(two[0], depth) and np.array_equal(two[1], conf)


def _sparse(shape, rows, cols, vals):
    a = np.zeros(tuple(shape))
    a[rows, cols] = vals
    return a


def test_generate_depth_map_from_kitti_files_vs_reference_golden(golden, tmp_path):
    """fusiondepth_amd.kitti_utils.generate_depth_map reading the reference's file formats == the reference's own
    generate_depth_map on the same files: default, shape=[384,1280], vel_depth=True, and the crop branch (shape=[352,1280])."""
    from fusiondepth_amd import kitti_utils as KU
    g = golden("rasterize_scan3")
    velo, P = gin.lidar_scan(int(g["seed"]))
    scan = _write_kitti_files(str(tmp_path), velo)
    P_file, im = KU.velo_to_image(str(tmp_path), 2)
    assert im == (375, 1242) and np.array_equal(P_file, P)
    assert KU.read_calib_file(os.path.join(str(tmp_path), "calib_cam_to_cam.txt"))["calib_time"] == "09-Jan-2012 13:57:47"
    full = _sparse((375, 1242), g["full_rows"], g["full_cols"], g["full_vals"])
    got = KU.generate_depth_map(str(tmp_path), scan, 2)
    assert got.dtype == np.float64 and np.array_equal(got, full)
    padded = KU.generate_depth_map(str(tmp_path), scan, 2, shape=[384, 1280])
    assert padded.shape == (384, 1280) and np.array_equal(padded[9:, 19:19 + 1242], full) and not padded[:9].any()
    assert np.array_equal(KU.generate_depth_map(str(tmp_path), scan, 2, True), _sparse((375, 1242), g["vd_rows"], g["vd_cols"], g["vd_vals"]))
    crop = KU.generate_depth_map(str(tmp_path), scan, 2, shape=[352, 1280])
    assert np.array_equal(crop, _sparse(g["crop_shape"], g["crop_rows"], g["crop_cols"], g["crop_vals"]))
    # get_4beam (gen2channel.py:42-58: metres, before the / 100) and its flipped twin
    pooled = KU.get_4beam(str(tmp_path), scan, 2, False)
    assert np.array_equal(pooled.astype(np.float32) / np.float32(100.0), g["beam"])
    assert np.array_equal(KU.get_4beam(str(tmp_path), scan, 2, True), np.fliplr(pooled))


def test_gen2channel_writes_the_npy_files_the_dataset_reads(golden, tmp_path):
    """gen2channel.py:122-183: <idx>_<side>_False.npy / _True.npy, float32 [2,192,640] == oracle scatter of the reference's
    4-beam map (and of its left-right flip)."""
    from fusiondepth_amd import kitti_utils as KU
    from oracle import scatter as OS
    g = golden("rasterize_scan3")
    velo, _ = gin.lidar_scan(int(g["seed"]))
    scan = _write_kitti_files(str(tmp_path), velo)
    out = os.path.join(str(tmp_path), "2channel")
    paths = KU.gen2channel(str(tmp_path), scan, out, 7, "l")
    assert [os.path.basename(p) for p in paths] == ["7_l_False.npy", "7_l_True.npy"]
    for p, beam in zip(paths, (g["beam"], np.ascontiguousarray(np.fliplr(g["beam"])))):
        two = np.load(p)
        assert two.dtype == np.float32 and two.shape == (2, 192, 640)
        d, c = OS.scatter_2channel_c(beam)
        assert np.array_equal(two[0], d) and np.array_equal(two[1], c)
    stamp = os.path.getmtime(paths[0])
    KU.gen2channel(str(tmp_path), scan, out, 7, "l", regenerate=False)          # both files exist: left alone
    assert os.path.getmtime(paths[0]) == stamp


@pytest.mark.parametrize("n_points", [100, 200])
def test_random_sample_scans_r100_r200(n_points):
    """BASELINE config 5's sparse inputs: `random100` / `random200` scans (gen2channel.py:17-24, --random_sample) hold 100 / 200
    Velodyne returns instead of 4 beams; the rasterise -> scatter chain is the same and must stay bit-exact on such maps."""
    import fusiondepth_amd.functional as FD
    from oracle import rasterize as OR
    from oracle import scatter as OS
    rng = np.random.RandomState(n_points)
    _, P = gin.lidar_scan(3)
    velo = np.stack([rng.uniform(4.0, 60.0, n_points), rng.uniform(-12.0, 12.0, n_points), rng.uniform(-1.6, 0.2, n_points),
                     rng.rand(n_points)], 1).astype(np.float32)
    beam = FD.velo_rasterize(torch.from_numpy(velo).cuda(), P, 375, 1242)
    want_beam = OR.four_beam(velo, P, 375, 1242)
    assert 0 < (want_beam > 0).sum() <= n_points
    assert np.array_equal(beam.cpu().numpy(), want_beam)
    two = FD.scatter_2channel(beam).cpu().numpy()
    d, c = OS.scatter_2channel_c(want_beam)
    assert np.array_equal(two[0], d) and np.array_equal(two[1], c)


def test_export_gt_depths_from_a_kitti_tree(tmp_path):
    """export_gt_depth.py: split lines -> vel_depth maps (float32) -> gt_depths.npz / r200.npz, on a synthetic KITTI tree."""
    from fusiondepth_amd import kitti_utils as KU
    from oracle import rasterize as OR
    root = str(tmp_path)
    date, drive = "2011_09_26", "2011_09_26/2011_09_26_drive_0002_sync"
    os.makedirs(os.path.join(root, drive, "velodyne_points/data"))
    os.makedirs(os.path.join(root, drive, "random200"))
    scans = {}
    for frame, seed in ((69, 31), (54, 32)):
        velo, P = gin.lidar_scan(seed, n_points=4000)
        scans[frame] = (velo, P)
        _write_kitti_files(os.path.join(root, date), velo)                 # calib files live in the date folder
        velo.tofile(os.path.join(root, drive, "velodyne_points/data", "%010d.bin" % frame))
        velo[:200].tofile(os.path.join(root, drive, "random200", "%010d.bin" % frame))
    lines = ["%s %010d l" % (drive, 69), "%s %010d l" % (drive, 54)]
    out = os.path.join(root, KU.split_output_name("eigen"))
    maps = KU.export_gt_depths(root, lines, "eigen", out)
    data = np.load(out)["data"]
    assert data.dtype == np.float32 and data.shape == (2, 375, 1242)
    for m, frame in zip(maps, (69, 54)):
        velo, P = scans[frame]
        want = OR.depth_image(velo, P, 375, 1242, vel_depth=True).astype(np.float32)
        assert np.array_equal(m, want)
    assert np.array_equal(data[1], maps[1])
    sparse = KU.export_gt_depths(root, lines[:1], "r200", os.path.join(root, KU.split_output_name("r200")))
    want = OR.depth_image(scans[69][0][:200], scans[69][1], 375, 1242, vel_depth=True).astype(np.float32)
    assert np.array_equal(sparse[0], want) and KU.split_output_name("r200") == "r200.npz" and KU.split_output_name("4beam") == "4beam.npz"
    with pytest.raises(ValueError):
        KU.export_gt_depths(root, lines, "eigen_benchmark")
