#!/usr/bin/env python3
"""Golden vectors for the parameter EMA, written by RUNNING the reference's toolkit/param_ema.py (build
container only; the reference's Python never travels).  A small model is stepped four times with
start_step = 1: the first step copies the parameters, the others average; the parameters of the
"training" model are perturbed between steps the way an optimizer would.

usage:  python tests/golden/make_golden_ema.py"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from toolkit.param_ema import ParameterEMA  # noqa: E402


def main():
    torch.manual_seed(41)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Linear(7, 1031), torch.nn.Linear(1031, 3, bias=False),
                                torch.nn.LayerNorm(9))
    ema_model = copy.deepcopy(model).eval().requires_grad_(False)
    with torch.no_grad():
        for p in ema_model.parameters():
            p.add_(1.0)                     # visibly different until the first step copies
    ema = ParameterEMA(start_step=1, beta=0.995)
    out = {}
    for step in range(4):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.05)
        for i, p in enumerate(model.parameters()):
            out["cur.%d.%d" % (step, i)] = p.detach().numpy().copy()
        ema.step_ema(ema_model, model, new_ema=True)
        for i, p in enumerate(ema_model.parameters()):
            out["ema.%d.%d" % (step, i)] = p.detach().numpy().copy()
    out["n_params"] = np.array(len(list(model.parameters())))
    np.savez_compressed(os.path.join(HERE, "ema_reference.npz"), **out)
    print("wrote ema_reference.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
