"""What a library batched fp32 GEMM reaches on the shapes a non-fused F(4x4, 3x3) would produce for the deep trunk layers
([36][M][C] x [36][C][T]) - the yardstick for a hand-written one: bmm_probe.py"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
for name, M, C, T in (("layer4 b24", 512, 512, 240), ("layer4 b12", 512, 512, 120), ("layer3 b24", 256, 256, 720), ("layer3 b12", 256, 256, 360),
                      ("layer2 b24", 128, 128, 2880), ("layer4 b24 T256", 512, 512, 256), ("layer3 b24 T768", 256, 256, 768)):
    U = torch.randn(36, M, C, device="cuda"); V = torch.randn(36, C, T, device="cuda")
    out = torch.empty(36, M, T, device="cuda")
    for _ in range(5): torch.bmm(U, V, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): torch.bmm(U, V, out=out)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e3 / 50
    fl = 2.0 * 36 * M * C * T
    print("%-18s [36][%d][%d] x [36][%d][%d]: %6.1f us  %5.1f TFLOP/s (%.2f of the fp32 MFMA peak)" % (name, M, C, C, T, t, fl / t / 1e6, fl / t / 1e6 / 157.3), flush=True)
