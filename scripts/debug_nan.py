import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12"])
tr = Trainer(opt)
mbs = [synthetic.make_batch(6, 192, 640, seed=1234 + i) for i in range(2)]
for step in range(4):
    for i, mb in enumerate(mbs):
        outputs, losses = tr.process_batch(mb)
        print(step, i, {k: float(v.detach()) for k, v in losses.items()})
        for s in range(4):
            d = outputs[("disp", s)]
            print("   disp%d min %.4g max %.4g nan %d" % (s, float(d.min()), float(d.max()), int(torch.isnan(d).sum())))
        (losses["loss"] / 2).backward()
    print("   grad nan:", int(torch.isnan(tr.flat.flat_grad).sum()), "grad absmax %.4g" % float(tr.flat.flat_grad.abs().max()))
    tr.optimizer_step()
    print("   param nan:", int(torch.isnan(tr.flat.flat_param).sum()))
