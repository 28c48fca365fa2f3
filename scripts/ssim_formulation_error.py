"""Numerical experiment (CPU, numpy float32 with float64-emulated FMAs) behind the SSIM formulation of csrc/photometric_ms.hip:
error of four ways to evaluate the SSIM loss map (layers.py:251-281) against a float64 evaluation and against the reference's
float32 arithmetic (oracle/layers.py).  On the 192x640 golden inputs: exact division by 9 -> 6.9e-5 max vs the float32 reference,
sums scaled by 81 -> 8.2e-5, sums of values centred at 0.5 and scaled by 81 (the kernel's form) -> 6.0e-5 vs the float32
reference but 3e-6 vs float64; the float32 reference itself is 6e-5 away from float64."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import inputs as gin
from oracle import layers as OL
f32 = np.float32
B,H,W = 1,192,640
inp, rng = gin.batch_inputs(505, B, H, W)
x = inp[("color", -1, 0)]; y = inp[("color", 0, 0)]
print("img range", float(x.min()), float(x.max()), float(x.mean()))
ref = OL.ssim(x, y).numpy()          # [B,3,H,W]
ref64 = OL.ssim(x.double(), y.double()).numpy()
xp = np.pad(x.numpy(), ((0,0),(0,0),(1,1),(1,1)), mode="reflect"); yp = np.pad(y.numpy(), ((0,0),(0,0),(1,1),(1,1)), mode="reflect")
def box(a):   # h then v sums in f32
    h = (a[..., :, :-2] + a[..., :, 1:-1]) + a[..., :, 2:]
    return (h[..., :-2, :] + h[..., 1:-1, :]) + h[..., 2:, :]
Sx, Sy, Sxx, Syy, Sxy = box(xp), box(yp), box(xp*xp), box(yp*yp), box(xp*yp)
C1, C2 = f32(0.01**2), f32(0.03**2)
def fin(n, d):
    return np.clip((f32(1) - n / d) * f32(0.5), 0, 1)
# variant 1: exact division by 9
mx, my = Sx / f32(9), Sy / f32(9)
sx, sy, sxy = Sxx / f32(9) - mx*mx, Syy / f32(9) - my*my, Sxy / f32(9) - mx*my
v1 = fin((2*mx*my + C1) * (2*sxy + C2), (mx*mx + my*my + C1) * (sx + sy + C2))
# variant 2: multiply by 1/9
r9 = f32(1.0/9.0)
mx, my = Sx * r9, Sy * r9
sx, sy, sxy = Sxx * r9 - mx*mx, Syy * r9 - my*my, Sxy * r9 - mx*my
v2 = fin((2*mx*my + C1) * (2*sxy + C2), (mx*mx + my*my + C1) * (sx + sy + C2))
# variant 3: scaled domain with fma emulated in f64
def fma(a, b, c): return (a.astype(np.float64) * np.float64(b) + np.asarray(c, np.float64)).astype(f32)
K1, K2 = f32(81 * 0.01**2), f32(81 * 0.03**2)
p = Sx * Sy
A1 = fma(p, 2.0, K1)
A2 = fma(p, -2.0, fma(Sxy, 18.0, K2))
q2 = fma(Sy, Sy, Sx * Sx)
B1 = q2 + K1
B2 = fma(Sxx + Syy, 9.0, K2) - q2
v3 = fin(A1 * A2, B1 * B2)
# variant 4: centered second moments (shift 0.5), scaled domain
xc, yc = xp - f32(0.5), yp - f32(0.5)
Cx, Cy, Cxx, Cyy, Cxy = box(xc), box(yc), box(xc*xc), box(yc*yc), box(xc*yc)
Sx4, Sy4 = Cx + f32(4.5), Cy + f32(4.5)
p = Sx4 * Sy4
A1 = fma(p, 2.0, K1)
A2 = fma(Cx*Cy, -2.0, fma(Cxy, 18.0, K2))
B1 = fma(Sy4, Sy4, Sx4*Sx4) + K1
B2 = fma(Cxx + Cyy, 9.0, K2) - fma(Cy, Cy, Cx*Cx)
v4 = fin(A1 * A2, B1 * B2)
for name, v in (("div9", v1), ("mul1/9", v2), ("scaled", v3), ("centered", v4), ("ref32", ref)):
    d = np.abs(v - ref); d64 = np.abs(v - ref64)
    print("%-9s vs ref32: max %.3g mean %.3g | vs f64: max %.3g mean %.3g | mean val %.6f (ref %.6f)" % (name, d.max(), d.mean(), d64.max(), d64.mean(), v.mean(), ref.mean()))
