"""Data-parallel step protocol (hint_amd/dp.py) with world_size 2 over gloo on CPU.

The GPU kernels cannot run here, so the per-rank compute is injected: the CPU oracle produces
the local flat gradient, a torch re-statement of the fused clamp+Adam kernel consumes it.  What
is under test is the host logic that the GPU trainer shares: contiguous equal row shards, ONE
all-reduce (sum) of the flat gradient arena, the 1/world scale applied before the +-5 clamp
(train_unconditional.py:128-129,140-141), identical parameters on every rank afterwards, and
equality with a single-process step on the global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hint_amd import dp
from oracle import hint_oracle as orc

D, WIDTHS, NBLOCKS, GLOBAL_B, STEPS = 6, [24, 12], 2, 64, 3


def _flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def _make_flow():
    flow = orc.OracleFlow(D, NBLOCKS, WIDTHS, seed=0, init_scale=None)
    for p in flow.parameters():
        p.requires_grad_(True)
    return flow


def _local_flat_grad(flow, x):
    for p in flow.parameters():
        p.grad = None
    z, J = flow.forward(x)
    l0, l1 = flow.loss_terms(z, J)
    (l0 + l1).backward()
    return _flat([p.grad for p in flow.parameters()])


def _adam_like_kernel(flow, state, flat_grad, scale, step, lr=3e-3, b1=0.9, b2=0.95, eps=1e-4, wd=1.86e-5, clamp=5.0):
    """same arithmetic as hint_adam_kernel (csrc/hint_optim.hip)"""
    with torch.no_grad():
        P = _flat(list(flow.parameters()))
        g = (flat_grad * scale).clamp(-clamp, clamp) + wd * P
        state["m"] = b1 * state["m"] + (1 - b1) * g
        state["v"] = b2 * state["v"] + (1 - b2) * g * g
        P = P - (lr / (1 - b1 ** step)) * state["m"] / (state["v"].sqrt() / (1 - b2 ** step) ** 0.5 + eps)
        off = 0
        for p in flow.parameters():
            p.copy_(P[off:off + p.numel()].view_as(p))
            off += p.numel()


def _run(rank, world, port, xs, out_q, buckets=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    flow = _make_flow()
    n = sum(p.numel() for p in flow.parameters())
    state = dict(m=torch.zeros(n), v=torch.zeros(n))
    for step, x in enumerate(xs, 1):
        lo, hi = dp.shard_rows(x.shape[0], *dp.world_info())
        # 50x gradient so that the +-5 clamp is active and its order w.r.t. the averaging matters
        if buckets == 1:
            dp.dp_step(lambda: 50.0 * _local_flat_grad(flow, x[lo:hi]),
                       lambda g, scale: _adam_like_kernel(flow, state, g, scale, step))
        else:       # the trainer's two buckets (second half of the blocks first): the same sums
            g = 50.0 * _local_flat_grad(flow, x[lo:hi])
            _adam_like_kernel(flow, state, g, dp.allreduce_buckets_(g, [n // 2]), step)
    out_q.put((rank, _flat([p.detach() for p in flow.parameters()]).numpy()))
    if world > 1:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_global_batch():
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(GLOBAL_B, D, generator=g) for _ in range(STEPS)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    _run(0, 1, 0, xs, q)                      # single process, global batch
    ref = q.get()[1]
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, xs, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(results[0], results[1])          # replicas stay identical
    np.testing.assert_allclose(results[0], ref, rtol=2e-5, atol=2e-7)   # == global-batch step
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, xs, q, 2)) for r in range(2)]
    for p in procs:
        p.start()
    two = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(two[0], results[0])              # two buckets == one bucket, bit for bit


def test_shard_rows_and_scale():
    assert dp.shard_rows(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        dp.shard_rows(100, 0, 8)
    g = torch.ones(5)
    assert dp.allreduce_sum_(g) == 1.0 and torch.equal(g, torch.ones(5))   # no process group: world 1
