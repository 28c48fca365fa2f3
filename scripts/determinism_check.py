import sys, torch
sys.path.insert(0, "/root/repo")
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
from fusiondepth_amd import synthetic
def run(mode, interleave):
    torch.manual_seed(99)
    opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"])
    tr = Trainer(opt, verbose=False)
    tr.interleave_encoders = interleave
    mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for mb in mbs:
        mb["_noise"] = [torch.randn(tr.batch_size, 2, 192, 640, device="cuda", generator=g) for _ in range(4)]
    for _ in range(4):
        (tr.train_step_graphed if mode == "graph" else tr.train_step)(mbs)
    torch.cuda.synchronize()
    return tr.flat.flat_param.double().sum().item(), tr.flat.flat_param.clone()
for mode in ("eager", "graph"):
    for il in (False, True):
        a = run(mode, il); b = run(mode, il)
        print(mode, "interleave", il, "equal:", bool(torch.equal(a[1], b[1])), a[0])
