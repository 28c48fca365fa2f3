"""Worker of tests/test_gpu_dp.py: one data-parallel rank of the HIP-backed Trainer.  Launched by torch.distributed.run with
FD_DIST_BACKEND=gloo so that two ranks can share the single GPU of the test box (RCCL refuses two ranks on one device; the
code path - broadcast of the initial state, flat-gradient all-reduce, 1/world folded into Adam - is the same)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from fusiondepth_amd import dp, synthetic
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    out_path, mode = sys.argv[1], sys.argv[2]            # mode: "same" (both ranks see one batch) | "split"
    rank, world, local_rank = dp.init_from_env()
    backend = dist.get_backend()
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if backend == "nccl" else 0)
    H, W, B = 64, 96, 2

    def opts():
        return MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", str(B),
                                         "--height", str(H), "--width", str(W)])

    def log(msg):
        print("[dp worker %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    log("init done")
    torch.manual_seed(100 + rank)                        # different initial weights per rank: the broadcast must fix that
    tr = Trainer(opts(), rank=rank, world_size=world, verbose=False)
    log("trainer built")
    init = tr.flat.flat_param.clone()
    t = init.clone()
    dist.broadcast(t, src=0)
    same_init = bool(torch.equal(t, init))

    def batch(step, r):
        b = synthetic.make_batch(B, H, W, seed=900 + step + (0 if mode == "same" else 31 * r))
        g = torch.Generator(device="cuda"); g.manual_seed(77 + step + (0 if mode == "same" else 13 * r))
        b["_noise"] = [torch.randn(B, 2, H, W, device="cuda", generator=g) for _ in range(4)]
        return b

    losses, overlapped = [], 0
    for step in range(2):
        losses.append(float(tr.train_step([batch(step, rank)])["loss"]))
        overlapped = tr.grad_sync.n_overlapped
        log("step %d done, loss %.6f, %d of %d buckets overlapped" % (step, losses[-1], overlapped, len(tr.grad_sync.buckets)))
    p = tr.flat.flat_param.clone()
    s = p.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    replicas_equal = bool(torch.equal(s, p * world))      # world = 2: x + x == 2x exactly
    res = {"rank": rank, "same_init": same_init, "replicas_equal": replicas_equal, "losses": losses,
           "finite": bool(torch.isfinite(p).all()), "backend": backend, "buckets_overlapped": overlapped, "buckets": len(tr.grad_sync.buckets),
           "param_checksum": float(p.double().sum()) + float((p.double() * p.double()).sum())}
    log("replica check done")
    if rank == 0:
        # single-process reference from the same initial state
        solo = Trainer(opts(), rank=0, world_size=1, verbose=False)
        with torch.no_grad():
            solo.flat.flat_param.copy_(init)
        from fusiondepth_amd import functional as FD
        FD.bump_weights_epoch(); FD.refresh_weight_layouts()
        for step in range(2):
            solo.train_step([batch(step, 0)])
        d = (solo.flat.flat_param - p).abs()
        res["solo_max_abs_diff"] = float(d.max())
        res["solo_rel_l2"] = float(d.norm() / solo.flat.flat_param.norm())
        res["moved"] = float((p - init).abs().max())
        # which tensors differ most (diagnostics when the comparison fails)
        worst = []
        names = [(k, n) for k, m in solo.models.items() for n, _ in m.named_parameters()]
        for (k, n), q, o in zip(names, solo.parameters_to_train, solo.flat.offsets):
            dd = float(d[o:o + q.numel()].max())
            if dd > 1e-6:
                worst.append(("%s.%s" % (k, n), dd))
        worst.sort(key=lambda t: -t[1])
        res["worst"] = worst[:12]
        res["n_bad_tensors"] = len(worst)
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
