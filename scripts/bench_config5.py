"""BASELINE.json config 5 on one GPU: the Refiner's optimiser step (refiner.py:264-297: frozen stage-1 networks, pseudo-3D refine
decoder, dense GDC target) at 640x192 and the Completor's (completor.py: dense completion at its 1216x352 resolution), synthetic
inputs resident in HBM, eager launch path, HIP-event-free wall clock bracketed by synchronize like bench.py.  One JSON line each.
    python scripts/bench_config5.py [--batch_size 12] [--steps 10] [--warmup 3]        (run on the GPU box)"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions

ap = argparse.ArgumentParser()
ap.add_argument("--batch_size", type=int, default=12)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()


def timed(step, n, w):
    for _ in range(w):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def refiner_line():
    from fusiondepth_amd.trainer import Trainer
    from fusiondepth_amd.refiner import Refiner
    base = ["--num_layers", "18", "--weights_init", "scratch", "--batch_size", str(args.batch_size), "--height", "192", "--width", "640"]
    folder = tempfile.mkdtemp(prefix="fd_stage1_")
    o1 = MonodepthOptions().parse(base + ["--log_dir", folder, "--model_name", "stage1"])
    tr = Trainer(o1, verbose=False)
    tr.save_model("stage1")
    w = os.path.join(tr.log_path, "models", "weights_stage1")
    del tr
    torch.cuda.empty_cache()
    rf = Refiner(MonodepthOptions().parse(base + ["--refine_load_weights_folder", w]), verbose=False)
    B = rf.batch_size
    inp = synthetic.make_batch(B, 192, 640, seed=77)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    inp["inf_gdc"] = torch.empty(B, 1, 192, 640, device="cuda").uniform_(0.05, 1.5, generator=gen)
    dt = timed(lambda: rf.train_step(inp), args.steps, args.warmup)
    loss = float(rf.train_step(inp)["loss"])
    n = sum(p.numel() for p in rf.parameters_to_train)
    return {"metric": "refiner training images/sec (640x192, ResNet-18 stage 1 frozen, refine2d decoder trained)", "value": B / dt,
            "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt,
            "higher_is_better": True, "dtype": "f32", "data": "synthetic", "final_loss": loss,
            "config": {"workload": "Refiner.train_step, --batch_size %d (one optimiser step per batch of %d, refiner.py:272-278), "
                                   "%d trained parameters" % (args.batch_size, B, n)}}


def completor_line():
    from fusiondepth_amd.completor import Completor
    o = MonodepthOptions().parse(["--weights_init", "scratch", "--batch_size", str(args.batch_size), "--completion_num_layers", "18"])
    cp = Completor(o, verbose=False)
    H, W = o.height, o.width
    mbs = [synthetic.make_batch(cp.batch_size, H, W, seed=31 + i) for i in range(cp.accumulate_step)]
    inp = cp.stack_micro_batches(mbs) if cp.stack_microbatches else mbs
    dt = timed(lambda: cp.train_step(inp), args.steps, args.warmup)
    loss = float(cp.train_step(inp)["loss"])
    imgs = cp.batch_size * cp.accumulate_step
    return {"metric": "completor training images/sec (%dx%d, ResNet-18)" % (W, H), "value": imgs / dt, "unit": "images/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "dtype": "f32",
            "data": "synthetic", "final_loss": loss,
            "config": {"workload": "Completor.train_step, --batch_size %d = %d micro-batch(es) of %d, %dx%d"
                                   % (args.batch_size, cp.accumulate_step, cp.batch_size, W, H)}}


for fn in (refiner_line, completor_line):
    try:
        print(json.dumps(fn()), flush=True)
    except Exception as e:                      # one workload failing must not hide the other's number
        import traceback
        traceback.print_exc()
        print(json.dumps({"metric": fn.__name__, "error": repr(e)}), flush=True)
    torch.cuda.empty_cache()
