"""Data-parallel Trainer on the GPU, world_size 2: broadcast of the initial state, all-reduce of the flat
gradient buffer, 1/world averaging inside the Adam kernel.  Two ranks share the one GPU of the test box over the gloo backend
(RCCL needs one device per rank; the 8-GPU RCCL run is the driver's)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, mode, extra_env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / ("dp_" + mode))
    env = dict(os.environ, FD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dp_gpu_worker.py"), out, mode]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    return [json.load(open(out + ".%d" % k)) for k in range(2)]


@pytest.mark.parametrize("interleave", ["0", "1"])
def test_two_ranks_same_batch_equal_single_process(tmp_path, interleave):
    """Both ranks see the same batch: mean of two identical gradients = the gradient, so two DP steps must land on the
    single-process parameters (up to the rounding of (g + g) / 2 inside Adam) and the replicas must stay bit-identical.
    Two processes contending for one GPU also make this a race detector: with the encoders issued in turns (interleave = 1) it
    caught a gradient tensor shared by two streams (functional._UpCat.backward)."""
    r0, r1 = _run(tmp_path, "same", {"FD_INTERLEAVE": interleave})
    for r in (r0, r1):
        assert r["same_init"] and r["replicas_equal"] and r["finite"]
    assert r0["losses"] == r1["losses"]
    assert r0["moved"] > 1e-5                                   # the optimiser did step
    assert r0["solo_max_abs_diff"] <= 1e-6 and r0["solo_rel_l2"] <= 1e-6, r0


def test_two_ranks_different_batches_stay_in_sync(tmp_path):
    """Each rank has its own shard of the global batch: different local losses, identical parameters after the exchange,
    and a different result than rank 0 training alone on its shard (the other rank's gradients did arrive)."""
    r0, r1 = _run(tmp_path, "split")
    for r in (r0, r1):
        assert r["same_init"] and r["replicas_equal"] and r["finite"]
    assert r0["losses"] != r1["losses"]
    assert r0["solo_max_abs_diff"] > 1e-6
