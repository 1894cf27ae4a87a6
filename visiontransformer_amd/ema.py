"""ParameterEMA of the reference's trainer (toolkit/param_ema.py:2-27) over the fused multi-tensor kernel.

Same constructor, attributes (`beta`, `start_step`, `step`) and methods.  The reference walks the
parameters in Python and issues three elementwise launches for each (old * beta, (1 - beta) * new, the
sum), rebinding `ema_params.data`; here all CUDA parameters of a dtype go through `dsw_ema_update` in one
launch per 128 tensors, updated in place, with bit-identical values.  Parameters that are not on the GPU
(or not dense fp32 / bf16) take the reference's expression.
"""
import ctypes

import torch

from . import _lib


def _launch(pairs, dev, dtype, beta):
    n = len(pairs)
    ema_p = (ctypes.c_void_p * n)(*[e.data_ptr() for e, _ in pairs])
    cur_p = (ctypes.c_void_p * n)(*[c.data_ptr() for _, c in pairs])
    numel = (ctypes.c_longlong * n)(*[e.numel() for e, _ in pairs])
    with torch.cuda.device(dev):
        _lib.call("dsw_ema_update", n, ctypes.cast(ema_p, ctypes.c_void_p), ctypes.cast(cur_p, ctypes.c_void_p),
                  ctypes.cast(numel, ctypes.c_void_p), _lib.dtype_code(dtype), float(beta), float(1 - beta), _lib.stream(dev))
    # the kernel wrote through raw pointers: tell torch, so that everything keyed on a tensor's version counter
    # (autograd's saved-tensor checks, the prepared-weight caches of functional.py / models.py) sees the change
    torch.autograd.graph.increment_version([e for e, _ in pairs])


def ema_update_(ema_tensors, cur_tensors, beta):
    """ema[i] <- ema[i] * beta + (1 - beta) * cur[i] for lists of dense CUDA tensors (pairs of equal shape
    and dtype, fp32 or bf16), in place, on the current stream."""
    groups = {}
    for e, c in zip(ema_tensors, cur_tensors):
        if e.shape != c.shape or e.dtype != c.dtype or e.device != c.device:
            raise RuntimeError("ema_update_: the tensors of a pair differ in shape, dtype or device")
        _lib.require_cuda(e, c)
        if not (e.is_contiguous() and c.is_contiguous()):
            raise RuntimeError("ema_update_: dense tensors only")
        groups.setdefault((e.device, e.dtype), []).append((e, c))
    for (dev, dtype), pairs in groups.items():
        _launch(pairs, dev, dtype, beta)


class ParameterEMA:
    def __init__(self, start_step=2000, beta=0.995):
        super().__init__()
        self.beta = beta
        self.start_step = start_step
        self.step = 0

    @staticmethod
    def _param_ids(modules):
        return [id(p) for m in modules for p in m._parameters.values()]

    def _pairs(self, ema_model, current_model):
        """(current, ema) parameter pairs of the two models.  Walking the module tree costs more than the
        update itself, so the flat lists are kept per model pair and rebuilt when a module of either model
        holds other Parameter objects than it did (same modules assumed)."""
        key = (id(ema_model), id(current_model))
        plan = self.__dict__.setdefault("_plans", {}).get(key)
        if plan is not None:
            mods, ids, pairs = plan
            if self._param_ids(mods) == ids:
                return pairs
        mods = list(ema_model.modules()) + list(current_model.modules())
        pairs = list(zip(current_model.parameters(), ema_model.parameters()))
        self._plans[key] = (mods, self._param_ids(mods), pairs)
        return pairs

    def update_model_average(self, ema_model, current_model):
        groups = {}
        for new, old in self._pairs(ema_model, current_model):
            dt = old.dtype
            if (old.is_cuda and new.is_cuda and dt == new.dtype and (dt == torch.float32 or dt == torch.bfloat16)
                    and old.shape == new.shape and old.device == new.device and old.is_contiguous() and new.is_contiguous()):
                key = (old.device, dt)
                g = groups.get(key)
                if g is None:
                    g = groups[key] = []
                g.append((old, new))
            else:
                old.data = self.update_average(old.data, new.data)
        for (dev, dt), pairs in groups.items():
            _launch(pairs, dev, dt, self.beta)

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new

    def step_ema(self, ema_model, model, new_ema=True):
        if self.step < self.start_step and new_ema:
            self.set_parameters(ema_model, model)
            self.step += 1
            return
        self.update_model_average(ema_model, model)
        self.step += 1

    def set_parameters(self, target_model, source_model):
        target_model.load_state_dict(source_model.state_dict())
