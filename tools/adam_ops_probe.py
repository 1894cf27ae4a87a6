"""Which arithmetic form do torch's multi-tensor kernels use on this GPU?  Each pass of torch's multi-tensor Adam is
run on its own and compared, bit for bit, with the candidate forms (fused multiply-add emulated through float64)."""
import torch

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
n = 1 << 20
m = torch.randn(n, device=dev, generator=gen)
g = torch.randn(n, device=dev, generator=gen) * 0.3
v = torch.rand(n, device=dev, generator=gen) * 0.1
p = torch.randn(n, device=dev, generator=gen)


def f32(x):
    return torch.tensor(x, dtype=torch.float32).item()      # the double a float scalar becomes


def fma(a, b, c):
    return (a.double() * b.double() + c.double()).float()


def report(name, got, cands):
    print(name, {k: int((got != c).sum()) for k, c in cands.items()})


w1 = 1 - 0.9
out = torch._foreach_lerp([m], [g], w1)[0]
d = g - m
w1t = torch.full_like(m, f32(w1))
report("lerp   ", out, {"fma(w,g-m,m)": fma(w1t, d, m), "m+rn(w*(g-m))": m + w1t * d})

w2 = 1 - 0.99
out = torch._foreach_addcmul([v], [g], [g], value=w2)[0]
w2t = torch.full_like(m, f32(w2))
report("addcmul", out, {"fma(rn(w*g),g,v)": fma(w2t * g, g, v), "rn(rn(w*g)*g)+v": (w2t * g) * g + v,
                        "fma(w,rn(g*g),v)": fma(w2t, g * g, v), "rn(w*rn(g*g))+v": w2t * (g * g) + v})

s = (1 - 0.99 ** 3) ** 0.5
sq = torch._foreach_sqrt([v])[0]
report("sqrt   ", sq, {"torch.sqrt": torch.sqrt(v), "f64 sqrt": v.double().sqrt().float()})
out = torch._foreach_div([sq], s)[0]
st = torch.full_like(m, f32(s))
report("div    ", out, {"x/s": sq / st, "f64 x/s": (sq.double() / st.double()).float(), "x*rn(1/s)": sq * (1.0 / st),
                        "x*f32(1/s_double)": sq * torch.full_like(m, f32(1.0 / s))})
den = out + 1e-8
a = (2e-4 / (1 - 0.9 ** 3)) * -1
out = torch._foreach_addcdiv([p], [m], [den], value=a)[0]
at = torch.full_like(m, f32(a))
q = (m.double() / den.double()).float()
report("addcdiv", out, {"fma(a,rn(m/d),p)": fma(at, q, p), "p+rn(a*rn(m/d))": p + at * q,
                        "p+rn(rn(a*m)/d)": p + ((at * m).double() / den.double()).float(),
                        "fma(a,m/d torch,p)": fma(at, m / den, p)})

# the reciprocal of the scalar divisor: float(1) / float(s), or the double 1 / s rounded to float?
import numpy as np
for t in range(1, 400):
    s = (1 - 0.99 ** t) ** 0.5
    r_float = float(np.float32(1.0) / np.float32(s))
    r_double = f32(1.0 / s)
    if r_float != r_double:
        out = torch._foreach_div([sq], s)[0]
        print("t=%d" % t, {"x*(1f/float(s))": int((out != sq * torch.full_like(m, r_float)).sum()),
                           "x*float(1/s)": int((out != sq * torch.full_like(m, r_double)).sum())})
        break

# torch/optim/adam.py passes LISTS of scalars (one per tensor) to the division and to addcdiv: other overloads
s = (1 - 0.99 ** 2) ** 0.5
st = torch.full_like(m, f32(s))
out = torch._foreach_div([sq], [s])[0]
report("div, scalar list    ", out, {"x/s": sq / st, "x*float(1/s)": sq * torch.full_like(m, f32(1.0 / s)),
                                     "f64 x/s_double": (sq.double() / s).float()})
out = torch._foreach_addcdiv([p], [m], [den], [a])[0]
report("addcdiv, scalar list", out, {"fma(a,rn(m/d),p)": fma(at, q, p), "p+rn(a*rn(m/d))": p + at * q,
                                     "p+rn(rn(a*m)/d)": p + ((at * m).double() / den.double()).float()})
