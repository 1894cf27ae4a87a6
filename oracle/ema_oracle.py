"""CPU restatement of the reference's parameter EMA (toolkit/param_ema.py:2-27).  Test infrastructure:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.  Pinned by
tests/golden/ema_reference.npz, written by tests/golden/make_golden_ema.py from a run of the reference's
own ParameterEMA."""
import torch


def update_average(old, new, beta):
    """toolkit/param_ema.py:14-17: three separately rounded elementwise ops in the tensors' dtype"""
    if old is None:
        return new
    return old * beta + (1 - beta) * new


def step_ema(state, ema_params, cur_params, new_ema=True):
    """toolkit/param_ema.py:19-27 on lists of tensors; state = dict(step, start_step, beta).  Returns the new
    list of EMA tensors."""
    if state["step"] < state["start_step"] and new_ema:
        out = [c.clone() for c in cur_params]           # set_parameters: load_state_dict copies the values
    else:
        out = [update_average(e, c, state["beta"]) for e, c in zip(ema_params, cur_params)]
    state["step"] += 1
    return out
