"""Parameters of a whole network drawn from a seed, tensor by tensor, keyed on the parameter's NAME.

A default-width network has 58 MB of parameters - too much to commit as a fixture - so the golden-vector
script (which fills the reference's network) and the GPU test (which fills this package's network) both call
this function: same names (the state_dict keys are the contract), same shapes, same values, independent of the
order in which either side constructs or initialises its modules.  Data generation only; no model code."""
import zlib

import torch


def fill_parameters(module, seed):
    with torch.no_grad():
        for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
            gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            if "rezero" in name:                                  # zero at init: every branch would be times 0
                v = torch.rand(p.shape, generator=gen) * 0.4 + 0.3
            elif name.endswith("relative_position_bias_table"):
                v = torch.randn(p.shape, generator=gen) * 0.5
            elif p.dim() >= 2:
                v = torch.randn(p.shape, generator=gen) / p[0].numel() ** 0.5
            elif name.endswith(".weight"):                        # norm gains
                v = 1.0 + torch.randn(p.shape, generator=gen) * 0.1
            else:
                v = torch.randn(p.shape, generator=gen) * 0.1
            if name.endswith("sparse_discriminator.4.weight"):    # gates on both sides of 0.5, offsets spread
                v = v * 6.0
            if name.endswith("sparse_discriminator.4.bias"):
                v[1] += 0.6
            p.copy_(v.to(p.dtype))
