"""The real FlowTrainer under data parallelism: two ranks sharing the one GPU of the test box, gloo
as the collective backend (RCCL refuses two ranks on one device; gloo stages CUDA tensors through the
host).  What is checked is what no CPU test can see: the HIP step on a row shard + ONE all-reduce of
the flat gradient arena + the fused clamp+Adam with grad_scale = 1/world reproduces the
single-process step on the global batch, and the replicas stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
D, WIDTHS, NBLOCKS, GLOBAL_B, STEPS = 6, [32, 16], 3, 512, 4


def _run(rank, world, port, xs, use_graph, out_q, buckets="2"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HINT_DP_BUCKETS"] = buckets
    import hint_amd
    from hint_amd import dp
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)                                   # identical initial weights on every rank
    flow = hint_amd.HintFlow(D, NBLOCKS, WIDTHS).to(dev)
    with torch.no_grad():
        for p in flow.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
    # 200x loss scale is not available in the trainer; a large learning rate and an active clamp come
    # from un-normalised inputs instead
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=use_graph, lr=3e-3)
    losses = []
    for x in xs:
        lo, hi = dp.shard_rows(x.shape[0], *dp.world_info())
        l0, l1 = tr.step((8.0 * x[lo:hi]).to(dev))
        losses.append([float(l0), float(l1)])
    flat = torch.cat([p.detach().reshape(-1) for p in flow.parameters()]).cpu().numpy()
    out_q.put((rank, flat, np.array(losses)))
    if world > 1:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_gpu_step_equals_single_process_global_batch(use_graph):
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(GLOBAL_B, D, generator=g) for _ in range(STEPS)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_run, args=(0, 1, 0, xs, use_graph, q))
    p0.start()
    _, ref, ref_losses = q.get(timeout=240)
    p0.join(timeout=60)
    assert p0.exitcode == 0
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, xs, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, flat, losses = q.get(timeout=240)
        res[r] = (flat, losses)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][0], res[1][0])                    # replicas stay identical
    np.testing.assert_allclose(res[0][0], ref, rtol=2e-4, atol=2e-6)       # == global-batch step
    # the mean of the two shards' loss terms is the global batch's
    np.testing.assert_allclose(0.5 * (res[0][1] + res[1][1]), ref_losses, rtol=1e-4, atol=1e-5)
    # the step above all-reduced the gradient arena in two buckets (part B of the last blocks, bucket, part B of the first
    # blocks, bucket): bit for bit the single-bucket step
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, xs, use_graph, q, "1")) for r in range(2)]
    for p in procs:
        p.start()
    one = {}
    for _ in range(2):
        r, flat, losses = q.get(timeout=240)
        one[r] = (flat, losses)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(one[0][0], res[0][0])
    np.testing.assert_array_equal(one[0][1], res[0][1])


def _run_cond(rank, world, port, xs, ys, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import hint_amd
    from hint_amd import dp
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = hint_amd.ConditionalHintFlow(10, 3, 2, 24).to(dev)
    with torch.no_grad():
        for p in model.parameters():
            p.data.add_(0.02 * torch.randn_like(p))
    tr = hint_amd.ConditionalFlowTrainer(model, noise=0.0, lr=3e-3)      # (gloo: plain launches, the all-reduce from the host)
    losses = []
    for x, y in zip(xs, ys):
        lo, hi = dp.shard_rows(x.shape[0], *dp.world_info())
        l0, l1 = tr.step((4.0 * x[lo:hi]).to(dev), y[lo:hi].to(dev))
        losses.append([float(l0), float(l1)])
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    out_q.put((rank, flat, np.array(losses)))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_conditional_step_equals_single_process_global_batch():
    """the conditional two-lane trainer (round 5: 25-launch step; with a process group part B, the all-reduce and the optimizer
    launch stay apart) on two ranks sharing the GPU over gloo: replicas bit-identical, equal to the one-process step on the
    global batch (train_conditional.py:120-150 on the union of the shards)"""
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(256, 10, generator=g) for _ in range(3)]
    ys = [torch.randn(256, 3, generator=g) for _ in range(3)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_run_cond, args=(0, 1, 0, xs, ys, q))
    p0.start()
    _, ref, ref_losses = q.get(timeout=240)
    p0.join(timeout=60)
    assert p0.exitcode == 0
    port = _free_port()
    procs = [ctx.Process(target=_run_cond, args=(r, 2, port, xs, ys, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, flat, losses = q.get(timeout=240)
        res[r] = (flat, losses)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][0], ref, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(0.5 * (res[0][1] + res[1][1]), ref_losses, rtol=1e-4, atol=1e-5)


def test_allreduce_captured_in_the_step_graph_and_host_side_fallback():
    """a one-rank RCCL group in this process: the gradient all-reduce and the optimizer are captured in
    the step's graph; when the collective cannot be captured (simulated) the trainer issues both from
    the host - same steps either way as a trainer without any process group"""
    import copy
    import hint_amd
    DEV = "cuda:0"
    torch.manual_seed(11)
    flow0 = hint_amd.HintFlow(6, 2, [32, 16]).to(DEV)
    for p in flow0.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    xs = [torch.randn(256, 6, device=DEV) for _ in range(3)]

    def run(flow, break_capture=False):
        tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=True)
        real = dist.all_reduce
        if break_capture:
            def flaky(t, *a, **k):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("simulated: collective not capturable")
                return real(t, *a, **k)
            dist.all_reduce = flaky
        try:
            out = []
            for x in xs:
                tr.step(x)
                out.append([float(v) for v in tr.last_losses()])
        finally:
            dist.all_reduce = real
        return tr, out

    cflow0 = hint_amd.ConditionalHintFlow(10, 3, 2, 24).to(DEV)
    for p in cflow0.parameters():
        p.data.add_(0.02 * torch.randn_like(p))
    cx = [torch.randn(128, 10, device=DEV) for _ in range(3)]
    cy = [torch.randn(128, 3, device=DEV) for _ in range(3)]

    def run_cond():
        tr = hint_amd.ConditionalFlowTrainer(copy.deepcopy(cflow0), noise=0.0, use_graph=True)
        out = []
        for x, y in zip(cx, cy):
            l0, l1 = tr.step(x, y)
            out.append([float(l0), float(l1)])
        return tr, out

    _, want = run(copy.deepcopy(flow0))                      # no process group at all
    _, cwant = run_cond()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tr1, got1 = run(copy.deepcopy(flow0))
        assert tr1._allreduce_in_graph and tr1._adam_in_graph
        assert tr1._split == 0 and "one all-reduce" in tr1.allreduce_plan()      # the default: ONE bucket, captured
        os.environ["HINT_DP_BUCKETS"] = "2"
        try:
            tr3, got3 = run(copy.deepcopy(flow0))
        finally:
            del os.environ["HINT_DP_BUCKETS"]
        assert tr3._allreduce_in_graph and tr3._adam_in_graph
        assert tr3._split == 1 and tr3._side is not None      # two buckets, the first one's all-reduce on the side stream, captured
        tr2, got2 = run(copy.deepcopy(flow0), break_capture=True)
        assert not tr2._allreduce_in_graph and not tr2._adam_in_graph
        ctr, cgot = run_cond()                               # the conditional two-lane trainer: same capture
        assert ctr._graph is not None
    finally:
        dist.destroy_process_group()
    assert np.allclose(got1, want, rtol=1e-5, atol=1e-6), (got1, want)
    assert np.allclose(got2, want, rtol=1e-5, atol=1e-6), (got2, want)
    assert np.allclose(got3, want, rtol=1e-5, atol=1e-6), (got3, want)
    assert np.allclose(cgot, cwant, rtol=1e-5, atol=1e-6), (cgot, cwant)
