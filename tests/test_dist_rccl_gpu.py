"""RCCL smoke on the one GPU a test box has: a process group of ONE rank over the nccl backend (= RCCL on ROCm), the
same flat [gradients | loss] all-reduce bench.py / dist.all_reduce_gradient issue with more ranks.  What it can show
here: the backend initialises on this image with the environment the ranks get, and the collective runs on the
device buffer in place.  The two-rank arithmetic of the wrapper is covered on CPU (tests/test_dist_gloo.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_all_reduce_on_device():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        grads = [torch.arange(12, dtype=torch.float32, device=dev).reshape(3, 4), torch.ones(5, device=dev)]
        flat = torch.cat([g.reshape(-1) for g in grads] + [torch.tensor([2.5], device=dev)])
        want = flat.clone()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)          # the collective of dist.all_reduce_gradient
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
    finally:
        dist.destroy_process_group()
