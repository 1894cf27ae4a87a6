"""The optimizer of the reference's trainer (toolkit/init.py:133: torch.optim.Adam(model.parameters(), lr,
betas=(0.9, 0.99)); stepped at toolkit/model_helper2.py:125) over the one-pass multi-tensor kernel.

`Adam` is a torch.optim.Adam: same constructor, param_groups, per-parameter state (`step`, `exp_avg`,
`exp_avg_sq`) and state_dict, so checkpoints written by either load into the other, and the reference's
learning-rate scheduler, GradScaler and clip_grad_norm_ calls work on it unchanged.  Only `step()` differs:
torch walks the tensor lists eight times (lerp, mul, addcmul, sqrt, div, add, addcdiv and the step counters);
here every dense fp32 CUDA parameter goes through `dsw_adam_update`, one pass and one launch per 64 tensors,
in the same order of operations and rounding (bit-identical to torch's GPU kernels).  Parameters the kernel does not take (CPU, other dtypes,
non-contiguous) take torch's own single-tensor update.

Options the reference does not use (weight_decay, amsgrad, maximize, capturable, differentiable, fused) are
refused at construction instead of being silently ignored.
"""
import ctypes

import torch
from torch.optim._functional import adam as _torch_adam

from . import _lib


def _scalars(step, lr, beta1, beta2, eps):
    """the six scalars of dsw_adam_update, formed in double as torch/optim/adam.py forms them"""
    bias2_sqrt = (1 - beta2 ** step) ** 0.5
    return (float(1 - beta1), float(beta2), float(1 - beta2), float(bias2_sqrt), float(eps),
            float((lr / (1 - beta1 ** step)) * -1))


def _launch(dev, params, grads, exp_avgs, exp_avg_sqs, scalars):
    n = len(params)
    tab = [(ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]) for ts in (params, grads, exp_avgs, exp_avg_sqs)]
    numel = (ctypes.c_longlong * n)(*[p.numel() for p in params])
    with torch.cuda.device(dev):
        _lib.call("dsw_adam_update", n, *[ctypes.cast(t, ctypes.c_void_p) for t in tab], ctypes.cast(numel, ctypes.c_void_p),
                  *scalars, _lib.stream(dev))
    # written through raw pointers: bump the version counters, which autograd's saved-tensor checks and the
    # prepared-weight caches of functional.py / models.py are keyed on
    torch.autograd.graph.increment_version(list(params) + list(exp_avgs) + list(exp_avg_sqs))


def adam_update_(params, grads, exp_avgs, exp_avg_sqs, *, step, lr, beta1, beta2, eps):
    """One Adam step (the `step`-th, counted from 1) on lists of dense fp32 CUDA tensors, in place, on the
    current stream.  All tensors of one call share the step number."""
    if step < 1:
        raise ValueError("adam_update_: step counts from 1")
    by_dev = {}
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        _lib.require_cuda(p, g, m, v)
        for t in (p, g, m, v):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != p.shape or t.device != p.device:
                raise RuntimeError("adam_update_: dense fp32 tensors of one shape and device per parameter")
        by_dev.setdefault(p.device, ([], [], [], []))
        for lst, t in zip(by_dev[p.device], (p, g, m, v)):
            lst.append(t)
    for dev, (ps, gs, ms, vs) in by_dev.items():
        _launch(dev, ps, gs, ms, vs, _scalars(step, lr, beta1, beta2, eps))


class _GroupPlan:
    """What one param group's step needs, kept between steps: the parameters the kernel takes with their moments,
    one CPU buffer that holds all their step counters (each state["step"] is a 0-dim view of it, so one add
    advances them all), the device pointers last seen (a changed pointer means somebody rebound `.data`:
    everything is checked again), and the ctypes tables built from them."""
    __slots__ = ("group_params", "n_params", "fast", "ms", "vs", "steps", "slow", "idle", "ptrs", "tables", "numel_tab", "dev",
                 "permuted")


def _table(ptrs):
    return ctypes.cast((ctypes.c_void_p * len(ptrs))(*ptrs), ctypes.c_void_p)


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        for name in ("maximize", "capturable", "differentiable", "fused", "decoupled_weight_decay"):
            if kwargs.get(name):
                raise ValueError("visiontransformer_amd.optim.Adam: %s is not supported" % name)
        if weight_decay != 0 or amsgrad:
            raise ValueError("visiontransformer_amd.optim.Adam: weight_decay and amsgrad are not supported "
                             "(the trainer uses neither)")
        if isinstance(lr, torch.Tensor):
            raise ValueError("visiontransformer_amd.optim.Adam: lr must be a Python number")
        # `foreach` (a choice between torch's own implementations) is kept in the param groups as given, so that a
        # checkpoint loaded into torch.optim.Adam leaves that choice as the trainer made it; step() does not read it
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, **kwargs)
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}                # the state tensors are new objects

    def state_dict(self):
        """As torch's, with the step counters as tensors of their own (inside the optimizer they are views of one
        buffer per group, which an optimizer that loads this dict must not go on sharing)."""
        sd = super().state_dict()
        sd["state"] = {k: {name: (val.clone() if name == "step" and isinstance(val, torch.Tensor) else val)
                           for name, val in st.items()} for k, st in sd["state"].items()}
        return sd

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans = {}

    def _check_group(self, group):
        if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"] or group.get("capturable") \
                or group.get("differentiable") or group.get("fused") or group.get("decoupled_weight_decay"):
            raise RuntimeError("visiontransformer_amd.optim.Adam: a param group asks for weight_decay, amsgrad, maximize, "
                               "capturable, differentiable, fused or decoupled_weight_decay; none is supported")
        # the kernel's first-moment update is fma(1 - beta1, g - m, m), the form torch's lerp takes for a weight below
        # 0.5; at 1 - beta1 >= 0.5 torch computes g - (g - m) * beta1 instead and the two are no longer bit-identical
        if not float(group["betas"][0]) > 0.5:
            raise RuntimeError("visiontransformer_amd.optim.Adam: beta1 = %r: the fused update follows torch bit for bit only "
                               "for beta1 > 0.5 (the reference trains with 0.9); use torch.optim.Adam for such a group"
                               % (group["betas"][0],))

    @staticmethod
    def _dense(p):
        """one element per address, no gaps: row-major, or channels-last for a 4-D / 5-D tensor (the update is
        elementwise, so any such layout will do as long as p, its moments and its gradient share it)"""
        return (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                or (p.dim() == 5 and p.is_contiguous(memory_format=torch.channels_last_3d)))

    @classmethod
    def _kernel_takes(cls, p, m, v, t):
        return (p.is_cuda and not t.is_cuda and p.layout is torch.strided and cls._dense(p)
                and all(x.dtype == torch.float32 and x.device == p.device and x.shape == p.shape and x.stride() == p.stride()
                        for x in (p, m, v)))

    def _build_plan(self, group):
        """Sorts the group's parameters into: taken by the kernel (one device), updated by torch's own single-tensor
        code (everything else that has a gradient or a state), and idle (neither yet: torch creates the state of a
        parameter at its first gradient, and so does this)."""
        plan = _GroupPlan()
        plan.group_params, plan.n_params = group["params"], len(group["params"])
        plan.fast, plan.ms, plan.vs, plan.slow, plan.idle, plan.dev = [], [], [], [], [], None
        step_dtype = torch.float64 if torch.get_default_dtype() == torch.float64 else torch.float32
        for p in group["params"]:
            state = self.state[p]
            if len(state) == 0:
                if p.grad is None:
                    plan.idle.append(p)
                    continue
                state["step"] = torch.tensor(0.0, dtype=step_dtype)          # as torch.optim.Adam._init_group
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v, t = state["exp_avg"], state["exp_avg_sq"], state["step"]
            if self._kernel_takes(p, m, v, t) and (plan.dev is None or plan.dev == p.device):
                plan.dev = p.device
                plan.fast.append(p)
                plan.ms.append(m)
                plan.vs.append(v)
            else:
                plan.slow.append(p)
        plan.steps = torch.tensor([float(self.state[p]["step"]) for p in plan.fast], dtype=step_dtype)
        for i, p in enumerate(plan.fast):
            self.state[p]["step"] = plan.steps[i]
        plan.ptrs = ([p.data_ptr() for p in plan.fast], [m.data_ptr() for m in plan.ms], [v.data_ptr() for v in plan.vs])
        plan.tables = tuple(_table(x) for x in plan.ptrs)
        plan.numel_tab = (ctypes.c_longlong * len(plan.fast))(*[p.numel() for p in plan.fast])
        plan.permuted = {i for i, p in enumerate(plan.fast) if not p.is_contiguous()}     # dense, but not row-major
        return plan

    def _plan_for(self, gi, group):
        plan = self._plans.get(gi)
        if plan is not None and plan.group_params is group["params"] and plan.n_params == len(group["params"]) \
                and not any(p.grad is not None for p in plan.idle):
            ptrs = ([p.data_ptr() for p in plan.fast], [m.data_ptr() for m in plan.ms], [v.data_ptr() for v in plan.vs])
            if ptrs == plan.ptrs:
                return plan
        plan = self._plans[gi] = self._build_plan(group)
        return plan

    def _launch(self, plan, idx, grads, t, group):
        """the kernel on the plan's parameters `idx` (None = all of them), which all make their t-th step"""
        beta1, beta2 = group["betas"]
        lr = float(group["lr"])
        if idx is None:
            n, tabs, numel, touched = len(plan.fast), plan.tables, plan.numel_tab, plan.fast
        else:
            n = len(idx)
            tabs = tuple(_table([col[i] for i in idx]) for col in plan.ptrs)
            numel = (ctypes.c_longlong * n)(*[plan.numel_tab[i] for i in idx])
            touched = [plan.fast[i] for i in idx]
        with torch.cuda.device(plan.dev):
            _lib.call("dsw_adam_update", n, tabs[0], _table([g.data_ptr() for g in grads]), tabs[1], tabs[2],
                      ctypes.cast(numel, ctypes.c_void_p), *_scalars(t, lr, beta1, beta2, group["eps"]), _lib.stream(plan.dev))
        # written through raw pointers: move the version counters (see _launch above)
        torch.autograd.graph.increment_version(touched)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            self._check_group(group)
            plan = self._plan_for(gi, group)
            if plan.fast:
                grads = [p.grad for p in plan.fast]
                idx = None
                if any(g is None for g in grads):
                    idx = [i for i, g in enumerate(grads) if g is not None]
                    grads = [grads[i] for i in idx]
                for j, g in enumerate(grads):
                    if g.layout is not torch.strided:
                        raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                    k = j if idx is None else idx[j]
                    if (k in plan.permuted or not g.is_contiguous()) and g.stride() != plan.fast[k].stride():
                        grads[j] = torch.empty_like(plan.fast[k]).copy_(g)      # read only: the parameter's layout, same values
                if idx is None:
                    plan.steps += 1
                    ts = plan.steps.tolist()
                elif idx:
                    sel = torch.tensor(idx, dtype=torch.long)
                    plan.steps[sel] += 1
                    ts = plan.steps[sel].tolist()
                else:
                    ts = []
                if ts and min(ts) == max(ts):
                    self._launch(plan, idx, grads, int(ts[0]), group)
                elif ts:                                 # parameters that joined later are at other step counts
                    where = idx if idx is not None else list(range(len(plan.fast)))
                    for t in sorted(set(ts)):
                        pick = [k for k, tk in enumerate(ts) if tk == t]
                        self._launch(plan, [where[k] for k in pick], [grads[k] for k in pick], int(t), group)
            rest = ([], [], [], [], [])
            for p in plan.slow:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                st = self.state[p]
                for lst, x in zip(rest, (p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"])):
                    lst.append(x)
            if rest[0]:
                beta1, beta2 = group["betas"]
                _torch_adam(rest[0], rest[1], rest[2], rest[3], [], rest[4], foreach=False, capturable=False,
                            differentiable=False, fused=False, has_complex=any(torch.is_complex(p) for p in rest[0]),
                            amsgrad=False, beta1=beta1, beta2=beta2, lr=float(group["lr"]), weight_decay=0, eps=group["eps"],
                            maximize=False)
        return loss
