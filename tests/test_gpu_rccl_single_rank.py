"""The Trainer's gradient exchange over the real RCCL backend with a one-rank communicator (see tests/rccl_single_rank_worker.py):
the only RCCL run a 1-GPU box allows.  Own process: the default process group must not leak into the other tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_trainer_exchange_over_real_rccl_one_rank_communicator(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl1.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FD_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_single_rank_worker.py"), out], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["plain_ok"] and res["finite"] and res["moved"] > 1e-5, res
    nb = res["buckets"]
    assert nb >= 6 and res["n_overlapped"] == [nb] * 3 and res["n_whole"] == [0] * 3, res      # every bucket left from inside backward
    assert res["work_types"] and all("Work" in t for t in res["work_types"]), res                # real asynchronous handles
    assert res["overlap_differs"] == 0 and res["whole_differs"] == 0, res
