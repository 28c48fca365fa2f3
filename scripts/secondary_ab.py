"""One secondary BASELINE configuration, timed like bench.py's other_configs (3 windows, median) - for same-box A/B runs of fd_tuning /
tuning.host variants set through the FD_* variables (read once at import by fusiondepth_amd.tuning).

    python scripts/secondary_ab.py r50|r18big|r18 [windows] [steps]
"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fusiondepth_amd import synthetic  # noqa: E402
from fusiondepth_amd.options import MonodepthOptions  # noqa: E402
from fusiondepth_amd.trainer import Trainer  # noqa: E402

CFG = {"r50": (50, 192, 640, 8), "r18big": (18, 320, 1024, 8), "r18": (18, 192, 640, 12)}


def main():
    layers, H, W, bs = CFG[sys.argv[1]]
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    torch.manual_seed(7)
    opt = MonodepthOptions().parse(["--num_layers", str(layers), "--weights_init", "scratch", "--batch_size", str(bs), "--height", str(H),
                                    "--width", str(W)])
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer(opt, verbose=False)
    pool = []
    for i in range(3):
        mbs = [synthetic.make_scene_batch(tr.batch_size, H, W, seed=4321 + 17 * i + j, clutter=0.5) for j in range(tr.accumulate_step)]
        for mb in mbs:
            mb.pop("depth_gt", None)
            for f in (-1, 1):
                mb.pop(("T_gt", f), None)
        pool.append(tr.stack_micro_batches(mbs))
    per_step = []
    for k in range(8):                       # the first steps one by one: where a cold start spends its time
        torch.cuda.synchronize(); t = time.perf_counter()
        tr.train_step(pool[k % 3])
        torch.cuda.synchronize(); per_step.append(1e3 * (time.perf_counter() - t))
    win = []
    for _ in range(windows):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(n):
            tr.train_step(pool[k % 3])
        torch.cuda.synchronize(); win.append(1e3 * (time.perf_counter() - t) / n)
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("FD_"))
    print("%-7s %-50s first steps %s | windows %s | median %.2f ms = %.1f images/s"
          % (sys.argv[1], tag or "(defaults)", " ".join("%.1f" % v for v in per_step), " ".join("%.2f" % w for w in win),
             sorted(win)[len(win) // 2], opt.batch_size / (1e-3 * sorted(win)[len(win) // 2])), flush=True)


if __name__ == "__main__":
    main()
