import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, "/root/repo")
def main():
    from fusiondepth_amd import dp, synthetic
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd import functional as FD
    rank, world, _ = dp.init_from_env()
    torch.cuda.set_device(0)
    H, W, B = 64, 96, 2
    opts = lambda: MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", str(B), "--height", str(H), "--width", str(W)])
    torch.manual_seed(100)
    tr = Trainer(opts(), rank=rank, world_size=world, verbose=False)
    init = tr.flat.flat_param.clone()
    def batch(step):
        b = synthetic.make_batch(B, H, W, seed=900 + step)
        g = torch.Generator(device="cuda"); g.manual_seed(77 + step)
        b["_noise"] = [torch.randn(B, 2, H, W, device="cuda", generator=g) for _ in range(4)]
        return b
    def grads_of(t, b):
        t.flat.zero_grad()
        outputs, losses = t.process_batch(b, groups=t.accumulate_step)
        losses["loss"].backward()
        t._join_side_streams()
        torch.cuda.synchronize()
        return t.flat.flat_grad.clone(), float(losses["loss"])
    res = []
    b0 = batch(0)
    saved = {k: {n: bb.clone() for n, bb in m.named_buffers()} for k, m in tr.models.items()}
    for rep in range(6):
        with torch.no_grad():
            for k, m in tr.models.items():
                for n, bb in m.named_buffers(): bb.copy_(saved[k][n])
        g, l = grads_of(tr, b0)
        if rep == 0: g0 = g
        d = (g - g0).abs()
        names = [(k, n) for k, m in tr.models.items() for n, _ in m.named_parameters()]
        bad = []
        for (k, n), q, o in zip(names, tr.parameters_to_train, tr.flat.offsets):
            dd = float(d[o:o + q.numel()].max())
            if dd > 0: bad.append(("%s.%s" % (k, n), dd, float(g0[o:o+q.numel()].abs().max())))
        res.append((l, len(bad), bad[:5]))
        dist.barrier()
    if True:
        print("rank", rank, json.dumps(res)[:1500], flush=True)
    dist.barrier(); dist.destroy_process_group()
main()
