"""Where does the HIP forward lose accuracy at full-size planes?  Encoder features and decoder outputs of the HIP modules and
of the float32 oracle against the float64 oracle, same weights, same input.  Metric: max |diff| / rms(ref) per tensor.
Usage: python scripts/layer_err_probe.py [H W B]   (development tool; tests/ holds the assertions)"""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import test_gpu_trainer as T

H, W, B = (int(a) for a in (sys.argv[1:4] + ["352", "1216", "1"][len(sys.argv) - 1:]))
opt = T._opts(height=H, width=W, batch_size=B)
tr, ot = T._make_pair(opt)
for m in list(tr.models.values()) + list(ot.models.values()):
    m.train()
m64 = T._float64_models(ot.models)
cin = ot.models["encoder"].state_dict()[[k for k in ot.models["encoder"].state_dict() if k.endswith("conv1.weight")][0]].shape[1]
x = torch.from_numpy(np.random.RandomState(5).rand(B, cin, H, W).astype(np.float32))


def err(a, r):
    a, r = a.detach().double().cpu(), r.detach().double().cpu()
    return float((a - r).abs().max() / r.pow(2).mean().sqrt()), float(((a - r).abs() / r.abs().clamp_min(1e-30)).max())


with torch.no_grad():
    f64 = m64["encoder"](x.double())
    f32 = ot.models["encoder"](x)
    fg = tr.models["encoder"](x.cuda())
    for i in range(5):
        print("feature %d %s  oracle32 %.2e  HIP %.2e   (max|d|/rms)" % (i, tuple(f64[i].shape), err(f32[i], f64[i])[0], err(fg[i], f64[i])[0]))
    # the trainer's own inputs: colour frame into the depth encoder, sparse LiDAR 2-channel map into the beam encoder
    inp, noise = T._batch(B, H, W, 900)
    col, two = inp[("color_aug", 0, 0)], inp["2channel"]
    for name, xin in (("encoder", col), ("beam_encoder", two)):
        if name not in ot.models:
            continue
        a64 = m64[name](xin.double()); a32 = ot.models[name](xin); ag = tr.models[name](xin.cuda())
        for i in range(len(a64)):
            print("%s(real input) feature %d  oracle32 %.2e  HIP %.2e" % (name, i, err(a32[i], a64[i])[0], err(ag[i], a64[i])[0]))
    if "beam_encoder" in ot.models:
        e64 = m64["encoder"](col.double()); b64 = m64["beam_encoder"](two.double())
        d64 = m64["depth"](e64, beam_features=b64)
        d32 = ot.models["depth"](ot.models["encoder"](col), beam_features=ot.models["beam_encoder"](two))
        dg = tr.models["depth"](tr.models["encoder"](col.cuda()), beam_features=tr.models["beam_encoder"](two.cuda()))
        for s in range(4):
            k = ("disp", s)
            print("real chain %s  oracle32 %.2e / %.2e   HIP %.2e / %.2e" % ((k,) + err(d32[k], d64[k]) + err(dg[k], d64[k])))
        ef = [f.float() for f in e64]; bf = [f.float() for f in b64]
        d32 = ot.models["depth"](ef, beam_features=bf)
        dg = tr.models["depth"]([f.cuda() for f in ef], beam_features=[f.cuda() for f in bf])
        for s in range(4):
            k = ("disp", s)
            print("real decoder alone %s  oracle32 %.2e / %.2e   HIP %.2e / %.2e" % ((k,) + err(d32[k], d64[k]) + err(dg[k], d64[k])))
    # decoder alone: float64 features (rounded to float32) into both
    fin = [f.float() for f in f64]
    d64 = m64["depth"]([f.double() for f in fin])
    d32 = ot.models["depth"](fin)
    dg = tr.models["depth"]([f.cuda() for f in fin])
    for s in range(4):
        k = ("disp", s)
        print("decoder alone %s  oracle32 %.2e / %.2e   HIP %.2e / %.2e   (max|d|/rms, max rel)" % ((k,) + err(d32[k], d64[k]) + err(dg[k], d64[k])))
    # whole chain
    d64 = m64["depth"](f64); d32 = ot.models["depth"](f32); dg = tr.models["depth"](fg)
    for s in range(4):
        k = ("disp", s)
        print("enc+dec %s  oracle32 %.2e / %.2e   HIP %.2e / %.2e" % ((k,) + err(d32[k], d64[k]) + err(dg[k], d64[k])))
