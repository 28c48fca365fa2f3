"""Print every loss term per optimiser step on the synthetic benchmark batch (finds where a NaN first appears).
usage: debug_nan.py [steps] [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mbs = [synthetic.make_batch(6, 192, 640, seed=1234 + i) for i in range(2)]
for seed in range(s0, s0 + ns):
    torch.manual_seed(seed)
    opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12"])
    tr = Trainer(opt, verbose=False)
    batch = tr.stack_micro_batches(mbs)
    for step in range(n):
        losses = tr.train_step(batch)
        vals = {str(k): float(v.detach()) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1}
        bad = [k for k, v in vals.items() if v != v]
        pn = int(torch.isnan(tr.flat.flat_param).sum())
        if pn or bad or step == n - 1:
            print("seed", seed, "step", step, "loss %.5f" % vals.get("loss", float("nan")), "nan terms:", bad[:8], "param nan:", pn)
            if pn or bad:
                outputs, l2 = tr.process_batch(dict(batch), groups=tr.accumulate_step)
                for s in range(4):
                    d = outputs[("disp", s)]
                    print("   disp%d min %.4g max %.4g nan %d" % (s, float(d.min()), float(d.max()), int(torch.isnan(d).sum())))
            break
