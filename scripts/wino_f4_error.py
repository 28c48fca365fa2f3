"""fp32 error of Winograd F(4x4, 3x3) against F(2x2, 3x3) and the direct convolution (numpy, CPU): the number the next round needs
before it builds an F(4x4, 3x3) kernel (known gaps).  Error = max |y - y64| / max |y64| over the output, y64 = float64 direct."""
import numpy as np
rng = np.random.RandomState(0)
# Lavin & Gray, F(4, 3): interpolation points 0, +-1, +-2, inf
G4 = np.array([[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]])
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)

def wino(x, w, G, BT, AT, m, dt):
    C, H, W = x.shape; M = w.shape[0]; a = m + 2
    G, BT, AT = G.astype(dt), BT.astype(dt), AT.astype(dt)
    U = np.einsum("ik,mckl,jl->ijmc", G, w.astype(dt), G).astype(dt)            # [a][a][M][C]
    xp = np.pad(x.astype(dt), ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((M, H, W), dt)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[:, ty:ty + a, tx:tx + a]
            V = np.einsum("ik,ckl,jl->ijc", BT, d, BT).astype(dt)
            Mx = np.einsum("ijmc,ijc->ijm", U, V).astype(dt)                     # fp32 accumulation over C
            y[:, ty:ty + m, tx:tx + m] = np.einsum("ik,klm,jl->mij", AT, Mx, AT).astype(dt)
    return y

def direct(x, w, dt):
    C, H, W = x.shape
    xp = np.pad(x.astype(dt), ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((w.shape[0], H, W), dt)
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("mc,chw->mhw", w[:, :, ky, kx].astype(dt), xp[:, ky:ky + H, kx:kx + W]).astype(dt)
    return y

for C, M, H, W, relu in ((64, 64, 16, 32, True), (128, 128, 16, 16, True), (256, 256, 8, 16, True), (512, 512, 4, 8, True), (64, 64, 16, 32, False)):
    x = rng.randn(C, H, W); x = np.maximum(x, 0) if relu else x                 # post-ReLU activations like the trunk's, or signed
    w = rng.randn(M, C, 3, 3) * np.sqrt(2.0 / (9 * C))
    y64 = direct(x, w, np.float64); sc = np.abs(y64).max()
    e = {"direct f32": np.abs(direct(x, w, np.float32) - y64).max() / sc,
         "F(2x2) f32": np.abs(wino(x, w, G2, BT2, AT2, 2, np.float32) - y64).max() / sc,
         "F(4x4) f32": np.abs(wino(x, w, G4, BT4, AT4, 4, np.float32) - y64).max() / sc}
    print("C %3d -> %3d  %2dx%2d %s   " % (C, M, H, W, "relu " if relu else "signed") + "   ".join("%s %.1e" % kv for kv in e.items()), flush=True)
