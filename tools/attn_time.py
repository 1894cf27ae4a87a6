#!/usr/bin/env python3
"""attention kernel times at the bench shapes (for variant libraries)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from visiontransformer_amd import _lib
dev = torch.device("cuda:0"); _lib.load()
out = {}
bench._attention_rooflines(dev, 20, 3, out)
print("fwd %.1f us  bwd %.1f us" % (out["win_attn_fwd"]["us"], out["win_attn_bwd"]["us"]), flush=True)
