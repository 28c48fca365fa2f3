"""Host mirror of the reference ``kitti_utils.py`` and of the LiDAR preprocessing around it, on the HIP kernels.

Same function names, arguments and return types as the reference (file formats included), so its dataset classes and its
offline ``gen2channel.py`` script can call these instead:

  * ``load_velodyne_points`` / ``read_calib_file``            kitti_utils.py:8-30   (on-disk formats: Velodyne ``.bin`` =
    float32 x 4 per point; ``calib_*.txt`` = ``key: v0 v1 ...`` lines)
  * ``generate_depth_map(calib_dir, velo_filename, cam, vel_depth, shape)``   kitti_utils.py:40-102  -> float64 numpy image
  * ``get_4beam`` / ``get_4beam_2channel`` / ``gen2channel``  kitti_dataset.py:93-117, gen2channel.py:42-58,60-117,122-183
    (writes ``<idx>_<side>_<flip>.npy`` float32 [2,192,640], the files ``KITTIDataset.load_4beam_2channel`` reads).

Parsing the two text files and the 3x4 matrix product stay on the host in float64 exactly as the reference computes them;
projection, z-buffer, padding, pooling and the 2-channel scatter run on the GPU (``fd_velo_rasterize``,
``fd_scatter_2channel``) and are bit-exact against the reference (tests/test_gpu_rasterize.py).  No CPU fallback.
"""
import os

import numpy as np
import torch

from . import functional as FD


def load_velodyne_points(filename):
    """Velodyne scan file (float32 x, y, z, reflectance per point) -> [N,4] float32 with the 4th column set to 1, i.e. points in
    homogeneous coordinates ready for the 3x4 projection (interface of kitti_utils.py:8-11)."""
    scan = np.fromfile(filename, dtype=np.float32)
    if scan.size % 4:
        raise ValueError("%s: %d floats is not a whole number of (x, y, z, reflectance) records" % (filename, scan.size))
    scan = scan.reshape(-1, 4)
    scan[:, 3] = 1.0
    return scan


def read_calib_file(path):
    """KITTI ``calib_*.txt`` -> {key: float64 vector | str}.  Every line is ``key: payload``; a payload whose tokens all parse
    as numbers becomes a float64 vector, anything else (e.g. ``calib_time: 09-Jan-2012 13:57:47``) stays the stripped string
    (interface of kitti_utils.py:14-30)."""
    table = {}
    with open(path, "r") as handle:
        for raw in handle:
            key, sep, payload = raw.partition(":")
            if not sep:
                continue
            payload = payload.strip()
            try:
                table[key] = np.asarray([float(tok) for tok in payload.split()], dtype=np.float64) if payload else payload
            except ValueError:
                table[key] = payload
    return table


def velo_to_image(calib_dir, cam=2):
    """Velodyne -> rectified image plane of camera ``cam``: (P [3,4] float64, (im_h, im_w)).  P = P_rect_0<cam> . R_rect_00 .
    Tr_velo_to_cam, multiplied left to right in float64 like kitti_utils.py:43-57 (the products are compared bit for bit with
    the reference's in tests/test_gpu_rasterize.py)."""
    intr = read_calib_file(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    extr = read_calib_file(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
    to_cam = np.eye(4)
    to_cam[:3, :3] = extr["R"].reshape(3, 3)
    to_cam[:3, 3] = extr["T"]
    rectify = np.eye(4)
    rectify[:3, :3] = intr["R_rect_00"].reshape(3, 3)
    projection = intr["P_rect_0%d" % cam].reshape(3, 4)
    width, height = (int(v) for v in intr["S_rect_02"][:2])
    return (projection @ rectify) @ to_cam, (height, width)


def _device_scan(velo_filename, device):
    return torch.from_numpy(load_velodyne_points(velo_filename)).to(device)


def generate_depth_map(calib_dir, velo_filename, cam=2, vel_depth=False, shape=None, device="cuda"):
    """kitti_utils.py:40-102."""
    P, (im_h, im_w) = velo_to_image(calib_dir, cam)
    full = FD.velo_rasterize(_device_scan(velo_filename, device), P, im_h, im_w, shape, return_full=True, vel_depth=vel_depth,
                             beam=False)
    return full.cpu().numpy()


def get_4beam_device(calib_dir, velo_filename, cam=2, do_flip=False, device="cuda"):
    """kitti_dataset.py:93-110 + mono_dataset.py:196-198 on the device: [192,640] float32, metres / 100."""
    P, (im_h, im_w) = velo_to_image(calib_dir, cam)
    beam = FD.velo_rasterize(_device_scan(velo_filename, device), P, im_h, im_w, (384, 1280))
    return torch.flip(beam, dims=[1]).contiguous() if do_flip else beam


def get_4beam(calib_dir, velo_filename, cam=2, do_flip=False, device="cuda"):
    """gen2channel.py:42-58: the pooled map in metres (float64 numpy, before the / 100)."""
    P, (im_h, im_w) = velo_to_image(calib_dir, cam)
    full = FD.velo_rasterize(_device_scan(velo_filename, device), P, im_h, im_w, (384, 1280), return_full=True, beam=False)
    pooled = torch.nn.functional.max_pool2d(full[None], 2, ceil_mode=True)[0].cpu().numpy()
    return np.fliplr(pooled) if do_flip else pooled


def get_4beam_2channel(fourbeam, height=192, width=640, expand=2):
    """gen2channel.py:60-117: [H,W] map (metres / 100) -> (expanded_depth, confidence_map), device tensors."""
    fourbeam = torch.as_tensor(fourbeam, dtype=torch.float32)
    if not fourbeam.is_cuda:
        fourbeam = fourbeam.cuda()
    two = FD.scatter_2channel(fourbeam.contiguous(), FD.scaled_roi(height, width), expand)
    return two[0], two[1]


def gen2channel(calib_dir, velo_filename, out_path, idx, side, regenerate=True, device="cuda"):
    """gen2channel.py:122-183 for one split line: writes ``<idx>_<side>_False.npy`` and ``<idx>_<side>_True.npy``
    (float32 [2,192,640]: expanded depth, confidence) under ``out_path``; returns the two paths."""
    side_map = {"2": 2, "3": 3, "l": 2, "r": 3}
    os.makedirs(out_path, exist_ok=True)
    paths = [os.path.join(out_path, "{}_{}_{}.npy".format(idx, side, flip)) for flip in (False, True)]
    if not regenerate and all(os.path.isfile(p) for p in paths):
        return paths
    for flip, path in zip((False, True), paths):
        beam = get_4beam_device(calib_dir, velo_filename, side_map[side], flip, device)
        np.save(path, FD.scatter_2channel(beam).cpu().numpy())
    return paths


# export_gt_depth.py:57-131: which scan of a split line feeds the ground-truth / sparse-input depth map
_SPLIT_SCAN_DIR = {"eigen": "velodyne_points/data", "demo_gt": "velodyne_points/data", "demo": "4beam", "r100": "random100",
                   "r200": "random200", "1beam": "1beam", "2beam": "2beam", "3beam": "3beam", "4beam": "4beam", "8beam": "8beam",
                   "16beam": "16beam"}
_SPLIT_OUTPUT = {"r100": "r100.npz", "r200": "r200.npz", "demo": "4beam.npz", "eigen": "gt_depths.npz", "demo_gt": "gt_depths.npz"}


def export_gt_depths(data_path, lines, split, output_path=None, device="cuda"):
    """export_gt_depth.py:29-131 for the Velodyne-based splits: one ``generate_depth_map(calib_dir, scan, 2, True)`` (vel_depth)
    per split line ``"<date>/<drive> <frame> <side>"``, float32, saved as ``np.savez_compressed(output, data=...)`` - the
    ``gt_depths.npz`` / ``4beam.npz`` / ``r100.npz`` files ``evaluate_depth.py`` loads.  Returns the list of maps.
    (``eigen_benchmark`` reads PNGs and the ``detec*`` splits need OpenCV to find the calibration: not covered.)"""
    if split not in _SPLIT_SCAN_DIR:
        raise ValueError("export_gt_depths: split %r is not Velodyne-based (supported: %s)" % (split, sorted(_SPLIT_SCAN_DIR)))
    gt_depths = []
    for line in lines:
        folder, frame_id, _ = line.split()
        calib_dir = os.path.join(data_path, folder.split("/")[0])
        velo_filename = os.path.join(data_path, folder, _SPLIT_SCAN_DIR[split], "{:010d}.bin".format(int(frame_id)))
        gt_depths.append(generate_depth_map(calib_dir, velo_filename, 2, True, device=device).astype(np.float32))
    if output_path is not None:
        same = all(g.shape == gt_depths[0].shape for g in gt_depths)
        data = np.array(gt_depths) if same else np.array(gt_depths, dtype=object)       # drives differ in image size
        np.savez_compressed(output_path, data=data)
    return gt_depths


def split_output_name(split):
    """export_gt_depth.py:116-127: file name the split's maps are saved under."""
    if split in ("1beam", "2beam", "3beam", "4beam", "8beam", "16beam"):
        return "{}.npz".format(split)
    return _SPLIT_OUTPUT.get(split, "gt_depths.npz")
