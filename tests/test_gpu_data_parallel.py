"""The data-parallel wrappers around the kernel-backed attention pair on one GPU: a one-rank RCCL group runs the code
N ranks run (hooks, the flat gradient buffer, the all-reduce on RCCL's stream) and must leave exactly the gradients of
the bare pair."""
import copy
import os
import socket
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_one_rank_rccl_group_leaves_the_gradients_of_the_bare_pair():
    import torch.distributed as dist
    import bench
    from visiontransformer_amd import parallel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        bare = bench.build_pair(dev)
        gen = torch.Generator(device=dev).manual_seed(11)
        x = torch.randn(4, bench.HW, bench.HW, bench.C, device=dev, generator=gen).requires_grad_()
        noise = -torch.empty(4, (bench.HW // bench.WIN) ** 2, device=dev).exponential_(generator=gen).log()
        wrapped = {
            "flat": parallel.FlatDataParallel(copy.deepcopy(bare)),
            "ddp": torch.nn.parallel.DistributedDataParallel(copy.deepcopy(bare), device_ids=[0], broadcast_buffers=False,
                                                             gradient_as_bucket_view=True),
        }
        for m in [bare] + [w.module for w in wrapped.values()]:
            m.sdwa.gumbel_noise = noise

        def step(model):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(x)
            y.float().sum().backward()
            inner = getattr(model, "module", model)
            g = [p.grad.clone() for p in inner.parameters()] + [x.grad.clone()]
            x.grad = None
            return g

        for it in range(3):          # the reducers change state after the first pass (bucket rebuild, grads that are views)
            want = step(bare)
            for name, w in wrapped.items():
                got = step(w)
                assert len(got) == len(want)
                for i, (a, b) in enumerate(zip(got, want)):
                    assert torch.equal(a, b), "%s, step %d, gradient %d differs" % (name, it, i)
            if it != 1:              # step 1 -> 2 keeps .grad: the next pass accumulates into the same tensors
                for m in [bare] + list(wrapped.values()):
                    m.zero_grad(set_to_none=True)
            else:
                for m in [bare] + list(wrapped.values()):
                    m.zero_grad(set_to_none=False)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
