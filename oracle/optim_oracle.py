"""CPU restatement of the optimizer step of the reference's trainer.  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may use it.

The reference creates torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.99)) at toolkit/init.py:133 and
steps it at toolkit/model_helper2.py:125 on CUDA parameters; the algorithm itself lives in torch (the
multi-tensor Adam of torch/optim/adam.py in the torch 2.10 this image holds), not in the reference's files.
This is that algorithm in numpy, one array at a time, fp32, in torch's order of operations and with the
roundings its GPU kernels have (lerp, addcmul and addcdiv end in one fused multiply-add; addcmul multiplies the
product of the two tensors by the scalar; the division by the scalar bias correction is a true division).  Pinned by tests/test_optim.py:
against torch.optim.Adam on the CPU to a stated tolerance (torch's CPU kernels round a few of these steps
differently), and through the kernel, which must equal this restatement AND torch.optim.Adam run on the GPU bit
for bit.
"""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """a * b + c with one rounding (through float64, where the product of two floats is exact)"""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def adam_step(p, g, m, v, *, step, lr, beta1, beta2, eps):
    """One step (numbered `step`, from 1) on fp32 arrays; returns the new (p, m, v).
    torch/optim/adam.py, _multi_tensor_adam: lerp, mul, addcmul, sqrt, div, add, addcdiv."""
    p, g, m, v = (np.asarray(x, dtype=f32) for x in (p, g, m, v))
    w1, b2, w2 = f32(1 - beta1), f32(beta2), f32(1 - beta2)
    bias2_sqrt = f32((1 - beta2 ** step) ** 0.5)
    step_size = f32((lr / (1 - beta1 ** step)) * -1)
    m = _fma(w1, (g - m).astype(f32), m)
    v = (v * b2).astype(f32)
    v = _fma(w2, (g * g).astype(f32), v)
    d = ((np.sqrt(v).astype(f32) / bias2_sqrt).astype(f32) + f32(eps)).astype(f32)
    p = _fma(step_size, (m / d).astype(f32), p)
    return p, m, v
