"""Worker of tests/test_gpu_rccl_single_rank.py: the data-parallel exchange of the Trainer over the REAL RCCL backend (torch
backend "nccl") with a communicator of ONE rank - all a 1-GPU box can offer.  An all-reduce over one rank moves no data between
devices, but everything else is the production path: ProcessGroupNCCL's communicator stream, the event it records on the caller's
current stream (the module stream the backward node runs on), asynchronous ``Work`` handles, ``Work.wait()`` ordering the optimiser
behind the collectives, the broadcast of the initial state.  tests/test_gpu_dp_stream_semantics.py checks the same contract against a
stub; this checks that the stub's contract is RCCL's.

The Trainer is built with world_size = 2 on that one-rank group: the sum over one rank is the gradient itself and Adam scales it by
1/2, so the reference run is the single-process Trainer whose optimiser is handed grad_scale = 1/2 as well - the parameters after three
steps must agree BIT FOR BIT (overlapped buckets and the single whole-buffer exchange)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    from fusiondepth_amd import synthetic, tuning
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    H, W, B = 64, 96, 2

    def opts():
        return MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", str(B),
                                         "--height", str(H), "--width", str(W)])

    def batches(n):
        out = []
        for i in range(n):
            b = synthetic.make_batch(B, H, W, seed=3100 + i)
            g = torch.Generator(device="cuda"); g.manual_seed(150 + i)
            b["_noise"] = [torch.randn(B, 2, H, W, device="cuda", generator=g) for _ in range(4)]
            out.append(b)
        return out

    def run(world, overlap, half_scale=False, steps=3):
        tuning.host.dp_overlap = overlap
        torch.manual_seed(4242)
        tr = Trainer(opts(), rank=0, world_size=world, verbose=False)
        if half_scale:                                     # the single-process reference: same 1/2 inside Adam, no exchange
            orig = tr.optimizer_step
            tr.optimizer_step = lambda grad_scale=1.0: orig(0.5 * grad_scale)
        init = tr.flat.flat_param.clone()
        n_over, kinds = [], set()
        for b in batches(steps):
            tr.train_step([b])
            n_over.append(tr.grad_sync.n_overlapped)
            kinds |= {type(h).__name__ for h in tr.grad_sync.handles}
        torch.cuda.synchronize()
        moved = float((tr.flat.flat_param - init).abs().max())
        return tr.flat.flat_param.clone(), n_over, len(tr.grad_sync.buckets), sorted(kinds), moved

    # a plain collective first: RCCL itself must work on this box (communicator of one rank)
    t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    plain_ok = bool(torch.equal(t, torch.arange(1 << 20, device="cuda", dtype=torch.float32)))

    solo, _, _, _, moved = run(1, True, half_scale=True)
    over, n_over, nb, kinds, _ = run(2, True)
    whole, n_whole, _, _, _ = run(2, False)
    res = {"backend": dist.get_backend(), "plain_ok": plain_ok, "buckets": nb, "n_overlapped": n_over, "n_whole": n_whole,
           "work_types": kinds, "finite": bool(torch.isfinite(solo).all()),
           "overlap_differs": int((solo != over).sum()), "whole_differs": int((solo != whole).sum()),
           "moved": moved}
    with open(out_path, "w") as fh:
        json.dump(res, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
