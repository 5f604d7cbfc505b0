"""The GELU of csrc/common.h (gelu_erf_f): v * sigmoid(v (a + b v^2 + c v^4)), argument clamped to |v| <= 9 — fit of (a, b, c) to
0.5 v (1 + erf(v / sqrt 2)) and the error of the fp32 evaluation (exp2 form, coefficients scaled by -log2 e) against fp64 erf."""
import math
import numpy as np
import torch
from scipy.optimize import minimize
vv = np.linspace(-9, 9, 400001)
ex = 0.5 * vv * (1 + np.vectorize(math.erf)(vv / math.sqrt(2)))
f3 = lambda p: np.max(np.abs(vv / (1 + np.exp(-(p[0] * vv + p[1] * vv ** 3 + p[2] * vv ** 5))) - ex))
r = minimize(f3, [1.595, 0.0740, -0.000703], method="Nelder-Mead", options=dict(xatol=1e-10, fatol=1e-12, maxiter=20000, maxfev=20000))
print("a, b, c =", r.x, " max |error| on [-9, 9] (fp64):", r.fun)
L = 1.4426950408889634
ka, kb, kc = (torch.tensor(np.float32(-L * x)) for x in r.x)
print("-log2(e) * (a, b, c) =", float(ka), float(kb), float(kc))
v = torch.linspace(-30, 30, 6000001, dtype=torch.float32)
vc = v.clamp(-9, 9)
t = vc * vc
g = v / (1 + torch.exp2(((kc * t + kb) * t + ka) * vc))
exact = 0.5 * v.double() * (1 + torch.erf(v.double() / math.sqrt(2)))
e = (g.double() - exact).abs()
print("fp32 evaluation: max |error|", e.max().item(), "at v =", v[e.argmax()].item(), "; beyond |v| = 9:", e[v.abs() > 9].max().item())
