"""Data-parallel Trainer on the GPU, world_size 2: broadcast of the initial state, bucketed all-reduce of the
flat gradient buffer issued from inside the backward pass, 1/world averaging inside the Adam kernel.  On a box with one GPU the
two ranks share it over the gloo backend (RCCL needs one device per rank); with two or more visible devices the same tests also
run over RCCL (backend "nccl"), one device per rank - that is what the driver's multi-GPU box exercises."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _backends():
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return ["gloo"] + (["nccl"] if n >= 2 else [])


def _run(tmp_path, mode, extra_env=None, backend="gloo"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / ("dp_" + mode + "_" + backend + "_" + "".join(sorted((extra_env or {}).values()))))
    env = dict(os.environ, FD_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dp_gpu_worker.py"), out, mode]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    return [json.load(open(out + ".%d" % k)) for k in range(2)]


@pytest.mark.parametrize("backend", _backends())
@pytest.mark.parametrize("interleave", ["0", "1"])
def test_two_ranks_same_batch_equal_single_process(tmp_path, interleave, backend):
    """Both ranks see the same batch: mean of two identical gradients = the gradient, so two DP steps must land on the
    single-process parameters (up to the rounding of (g + g) / 2 inside Adam) and the replicas must stay bit-identical.
    Two processes contending for one GPU also make this a race detector: with the encoders issued in turns (interleave = 1) it
    caught a gradient tensor shared by two streams (functional._UpCat.backward)."""
    r0, r1 = _run(tmp_path, "same", {"FD_INTERLEAVE": interleave}, backend)
    for r in (r0, r1):
        assert r["same_init"] and r["replicas_equal"] and r["finite"] and r["backend"] == backend
        assert r["buckets_overlapped"] == r["buckets"] >= 6, r      # every bucket left from inside the backward pass
    assert r0["losses"] == r1["losses"]
    assert r0["moved"] > 1e-5                                   # the optimiser did step
    assert r0["solo_max_abs_diff"] <= 1e-6 and r0["solo_rel_l2"] <= 1e-6, r0


@pytest.mark.parametrize("backend", _backends())
def test_two_ranks_different_batches_stay_in_sync(tmp_path, backend):
    """Each rank has its own shard of the global batch: different local losses, identical parameters after the exchange,
    and a different result than rank 0 training alone on its shard (the other rank's gradients did arrive).  The overlapped
    bucket exchange and the single whole-buffer all-reduce after the backward pass must give the same parameters bit for bit
    (a sum of two addends does not depend on how the buffer is cut)."""
    r0, r1 = _run(tmp_path, "split", None, backend)
    for r in (r0, r1):
        assert r["same_init"] and r["replicas_equal"] and r["finite"]
    assert r0["losses"] != r1["losses"]
    assert r0["solo_max_abs_diff"] > 1e-6
    q0, q1 = _run(tmp_path, "split", {"FD_DP_OVERLAP": "0"}, backend)
    assert q0["buckets_overlapped"] == 0 and r0["buckets_overlapped"] > 0
    assert q0["param_checksum"] == r0["param_checksum"] and q0["losses"] == r0["losses"]


@pytest.mark.parametrize("backend", _backends())
def test_bench_two_ranks_reports_the_exchange(tmp_path, backend):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per GPU; on a one-GPU box two ranks share the
    device over gloo): ONE JSON line from rank 0 that carries the rank count, the all-reduce time and the overlap estimate."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FD_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--height", "64", "--width", "96", "--batch_size", "2", "--no_roofline", "--no_cpu_baseline"]
    if backend == "gloo":
        cmd.append("--share_device")         # two ranks on the one GPU of the test box: allowed only when asked for
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["backend"] == backend and res["config"]["parallelism"] == "dp2"
    assert res["allreduce_ms"] > 0 and 0.0 <= res["overlap_frac"] <= 1.0 and res["buckets_overlapped"] == res["allreduce_buckets"]
    assert res["value"] > 0 and res["params_finite"]


def _bench_plain(extra, timeout=500, lean=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FD_DIST_BACKEND", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "64",
           "--width", "96", "--batch_size", "2"] + (["--no_roofline", "--no_cpu_baseline"] if lean else []) + extra
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_plain_bench_gpus2_launches_its_own_ranks():
    """``python bench.py --gpus 2`` WITHOUT torch.distributed.run - the way the driver's single-GPU harness invokes bench.py -
    must start two ranks itself.  With two or more devices: one rank per device over RCCL, no flag.
    On the one-GPU test box the same command must REFUSE (it cannot measure two GPUs there) and name the reason; with
    --share_device the two ranks share the device over gloo, which exercises the self-launch path end to end."""
    import torch
    if torch.cuda.device_count() >= 2:
        r = _bench_plain([])
        want_backend = "nccl"
    else:
        refused = _bench_plain([], timeout=120)
        assert refused.returncode != 0 and "only 1 device" in refused.stderr and not [l for l in refused.stdout.splitlines() if l.startswith("{")]
        r = _bench_plain(["--share_device"])
        want_backend = "gloo"
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["backend"] == want_backend and res["config"]["parallelism"] == "dp2"
    assert res["final_loss"] is not None and res["params_finite"] and res["value"] > 0


def test_bench_gpus2_line_carries_roofline_and_cpu_baseline():
    """the N > 1 line must be complete the first time a multi-GPU node runs it - `roofline` (rank 0's probes)
    and `cpu_baseline` (rank 0, after the process group is gone so that no collective waits on a CPU run) next to the exchange figures."""
    import torch
    r = _bench_plain([] if torch.cuda.device_count() >= 2 else ["--share_device"], timeout=900, lean=False)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2
    for key in ("roofline", "roofline_loss_path", "cpu_baseline", "cpu_baseline_1thread", "allreduce_ms", "overlap_frac"):
        assert key in res, (key, sorted(res))
    assert res["roofline"]["bound"] == "mfma" and res["roofline"]["frac"] > 0 and res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["kind"] == "port"
