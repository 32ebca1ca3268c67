"""N>1 path on CPU: world_size-2 gloo processes exercise the shot partition and the single
all-reduce of [gradient | loss] that the RCCL path performs on GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shot_partition_covers_all_shots_once():
    from physicsbasedfwi2_amd.dist import shot_partition
    for S in (1, 5, 29, 32, 35, 69, 256):
        for R in (1, 2, 3, 4, 8):
            seen = []
            for r in range(R):
                lo, hi = shot_partition(S, r, R)
                assert 0 <= lo <= hi <= S
                seen += list(range(lo, hi))
            assert seen == list(range(S))
    assert shot_partition(29, 7, 8) == (28, 29)       # 29 shots / 8 GPUs: last rank gets one


def test_balanced_partition_for_strong_scaling():
    """`bench.py --scaling strong`: contiguous blocks whose sizes differ by at most one - C2's 29 shots on 8 ranks are
    4,4,4,4,4,3,3,3 - covering every shot once."""
    from physicsbasedfwi2_amd.dist import shot_partition_balanced
    sizes = [hi - lo for lo, hi in (shot_partition_balanced(29, r, 8) for r in range(8))]
    assert sizes == [4, 4, 4, 4, 4, 3, 3, 3]
    for S in (1, 5, 29, 32, 35, 69, 256):
        for R in (1, 2, 3, 4, 8):
            seen, sz = [], []
            for r in range(R):
                lo, hi = shot_partition_balanced(S, r, R)
                seen += list(range(lo, hi))
                sz.append(hi - lo)
            assert seen == list(range(S)) and max(sz) - min(sz) <= 1 and sz == sorted(sz, reverse=True)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from physicsbasedfwi2_amd.dist import all_reduce_gradient, shot_partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # per-shot "gradients": shot s contributes (s+1) * pattern; loss s contributes s+1
    S = 7
    lo, hi = shot_partition(S, rank, world)
    g1 = torch.zeros(3, 4)
    g2 = torch.zeros(5)
    loss = 0.0
    for s in range(lo, hi):
        g1 += (s + 1) * torch.arange(12.0).view(3, 4)
        g2 += (s + 1) * torch.ones(5)
        loss += float(s + 1)
    (g1, g2), total = all_reduce_gradient([g1, g2], loss)
    q.put((rank, g1.numpy(), g2.numpy(), float(total)))
    dist.destroy_process_group()


def test_all_reduce_gradient_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tot = sum(range(1, 8))
    for _, g1, g2, loss in res:
        assert np.allclose(g1, tot * np.arange(12.0).reshape(3, 4))
        assert np.allclose(g2, tot)
        assert loss == pytest.approx(tot)


def test_all_reduce_is_identity_without_process_group():
    from physicsbasedfwi2_amd.dist import all_reduce_gradient
    g = torch.ones(4)
    (out,), loss = all_reduce_gradient([g], 2.0)
    assert out is g and loss == 2.0
