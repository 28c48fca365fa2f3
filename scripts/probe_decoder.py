"""DepthDecoder forward + backward alone (batch 12, 192x640 feature pyramid), for rocprofv3: the step's serial section."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from fusiondepth_amd import networks, functional as FD
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
H, W = 192, 640
ch = [64, 64, 128, 256, 512]
dec = networks.DepthDecoder(np.array(ch)).cuda()
FD.enable_weight_cache(dec.parameters())
feats = [torch.randn(B, c, H >> (i + 1), W >> (i + 1), device="cuda", requires_grad=True) for i, c in enumerate(ch)]
beam = [torch.randn_like(f) for f in feats]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    out = dec(feats, beam_features=beam)
    loss = sum(out[("disp", s)].sum() for s in range(4))
    loss.backward()
torch.cuda.synchronize(); print("done")
