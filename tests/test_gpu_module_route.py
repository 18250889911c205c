"""GPU tests of the drop-in MODULE route (nn.Module + autograd, the path train_unconditional.py:114-144 takes): the fused
chained launches of HintFlow, the block-by-block walk, 'direct' parameter gradients against gradients through autograd, the
accumulation rules of p.grad, pooled tapes under retained graphs and overlapping forwards, the opt-in re-pack cache."""
import numpy as np
import pytest
import torch

import hint_amd
from hint_amd import hint as H
from util import CHAIN_CASES, load_chain_case, rel_err
from test_gpu_flow import build_flow, check_update

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_flow(d=6, nb=3, widths=(32, 16), dc=0, perm_first=False, reshuffle=False, seed=3):
    torch.manual_seed(seed)
    flow = hint_amd.HintFlow(d, nb, list(widths), ndim_c=dc, perm_first=perm_first, reshuffle=reshuffle).to(DEV)
    for p in flow.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    return flow


def loss_of(flow, x, c=None):
    z = flow(x, c=c)
    J = flow.log_jacobian(run_forward=False)
    return 0.5 * torch.sum(z ** 2, dim=1).mean() - J.mean()


def grads_of(flow, x, c=None, fuse=True, mode="direct"):
    prev = H.set_param_grad_mode(mode)
    try:
        flow.fuse_chain = fuse
        for p in flow.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        cg = c.clone().requires_grad_(True) if c is not None else None
        L = loss_of(flow, xg, cg)
        L.backward()
        out = [p.grad.detach().clone() for p in flow.parameters()]
        return float(L), xg.grad.clone(), (cg.grad.clone() if cg is not None else None), out
    finally:
        H.set_param_grad_mode(prev)
        flow.fuse_chain = True


@pytest.mark.parametrize("d,nb,widths,dc,perm_first,reshuffle,B", [
    (6, 3, (32, 16), 0, False, False, 333),
    (8, 4, (64, 32, 16), 3, True, False, 1000),
    (7, 2, (24, 12, 6), 0, False, True, 200),
    (21, 2, (48, 40, 24, 16), 2, False, False, 4100),
])
def test_fused_route_equals_block_walk_and_autograd_mode(d, nb, widths, dc, perm_first, reshuffle, B):
    """one flow, three routes: chained launches + direct gradients (the default), block by block + direct gradients, block by
    block with every parameter an autograd input (rounds 1-5): loss, dL/dx, dL/dc and every p.grad agree"""
    flow = make_flow(d, nb, widths, dc, perm_first, reshuffle)
    x = torch.randn(B, d, device=DEV)
    c = torch.randn(B, dc, device=DEV) if dc else None
    ref = grads_of(flow, x, c, fuse=False, mode="autograd")
    for fuse, mode in ((True, "direct"), (False, "direct")):
        got = grads_of(flow, x, c, fuse=fuse, mode=mode)
        assert abs(got[0] - ref[0]) <= 1e-5 * abs(ref[0]) + 1e-6
        assert rel_err(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 1e-4
        if dc:
            assert rel_err(got[2].cpu().numpy(), ref[2].cpu().numpy()) < 1e-4
        for g, r in zip(got[3], ref[3]):
            assert rel_err(g.cpu().numpy(), r.cpu().numpy()) < 1e-4
    # sampling direction and plain evaluation: fused launches against the walk
    with torch.no_grad():
        for rev in (False, True):
            flow.fuse_chain = True
            a = flow(x, c=c, rev=rev); Ja = flow.log_jacobian(run_forward=False)
            flow.fuse_chain = False
            b = flow(x, c=c, rev=rev); Jb = flow.log_jacobian(run_forward=False)
            flow.fuse_chain = True
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(Ja.cpu().numpy(), Jb.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("fuse", [True, False])
def test_direct_gradients_follow_autograd_accumulation_rules(fuse):
    """p.grad None -> set; left from an earlier backward -> added to; a foreign tensor / a mix -> added to in place"""
    flow = make_flow()
    flow.fuse_chain = fuse
    x = torch.randn(257, 6, device=DEV)
    ps = list(flow.parameters())
    loss_of(flow, x).backward()
    g1 = [p.grad.clone() for p in ps]
    loss_of(flow, x).backward()                       # no zero_grad: accumulates
    for p, g in zip(ps, g1):
        assert rel_err(p.grad.cpu().numpy(), (2 * g).cpu().numpy()) < 1e-6
    for i, p in enumerate(ps):                        # a mix: foreign tensors, None, and the views of the backward before
        if i % 3 == 0:
            p.grad = torch.ones_like(p)
        elif i % 3 == 1:
            p.grad = None
    held = [p.grad for p in ps]
    loss_of(flow, x).backward()
    for i, (p, g) in enumerate(zip(ps, g1)):
        want = g + 1 if i % 3 == 0 else (g if i % 3 == 1 else 3 * g)
        assert rel_err(p.grad.cpu().numpy(), want.cpu().numpy()) < 1e-6, i
        if i % 3 == 0:
            assert p.grad is held[i]                  # added to in place, like AccumulateGrad
    # optimizer protocol: zero_grad (set_to_none) -> backward -> clamp -> step, twice
    opt = torch.optim.Adam(ps, lr=1e-3)
    for _ in range(2):
        opt.zero_grad()
        loss_of(flow, x).backward()
        for p in ps:
            p.grad.data.clamp_(-5.0, 5.0)
        opt.step()
    opt.zero_grad(set_to_none=False)                  # zeroed in place: the next backward adds to zeros
    loss_of(flow, x).backward()
    ref = grads_of(flow, x, fuse=False, mode="autograd")[3]
    for p, r in zip(ps, ref):
        assert rel_err(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("fuse", [True, False])
def test_pooled_tapes_survive_retained_graphs_and_overlapping_forwards(fuse):
    flow = make_flow(d=8, nb=2, widths=(48, 24, 12))
    flow.fuse_chain = fuse
    ps = list(flow.parameters())
    xa, xb = torch.randn(300, 8, device=DEV), torch.randn(300, 8, device=DEV)
    ga = grads_of(flow, xa, fuse=fuse)[3]
    gb = grads_of(flow, xb, fuse=fuse)[3]
    flow.fuse_chain = fuse
    # two forwards alive at once, backward in the other order
    for p in ps:
        p.grad = None
    La, Lb = loss_of(flow, xa), loss_of(flow, xb)
    Lb.backward()
    for p, g in zip(ps, gb):
        assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < 1e-6
    for p in ps:
        p.grad = None
    La.backward()
    for p, g in zip(ps, ga):
        assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < 1e-6
    # retain_graph: the same node runs twice, with another forward in between
    for p in ps:
        p.grad = None
    La = loss_of(flow, xa)
    La.backward(retain_graph=True)
    with torch.no_grad():
        flow(xb)
    loss_of(flow, xb)                                  # (dropped without a backward: its tape goes back to the pool)
    for p in ps:
        p.grad = None
    La.backward()
    for p, g in zip(ps, ga):
        assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < 1e-6


def test_autograd_mode_serves_the_functional_api_and_hooks():
    flow = make_flow()
    x = torch.randn(100, 6, device=DEV)
    ref = grads_of(flow, x, fuse=False, mode="autograd")[3]
    prev = H.set_param_grad_mode("autograd")
    try:
        ps = list(flow.parameters())
        fired = []
        h = ps[0].register_hook(lambda g: fired.append(1))
        gs = torch.autograd.grad(loss_of(flow, x), ps)
        h.remove()
        assert fired
        for g, r in zip(gs, ref):
            assert rel_err(g.cpu().numpy(), r.cpu().numpy()) < 1e-6
    finally:
        H.set_param_grad_mode(prev)


def test_partly_frozen_flow_takes_the_autograd_route():
    flow = make_flow()
    x = torch.randn(100, 6, device=DEV)
    ref = grads_of(flow, x, fuse=False, mode="autograd")[3]
    ps = list(flow.parameters())
    for p in ps[:7]:
        p.requires_grad_(False)
        p.grad = None
    for p in ps:
        p.grad = None
    loss_of(flow, x).backward()
    for i, (p, r) in enumerate(zip(ps, ref)):
        if i < 7:
            assert p.grad is None
        else:
            assert rel_err(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-6
    for p in ps:                                       # nothing trainable: gradient of the input alone
        p.requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    loss_of(flow, xg).backward()
    assert xg.grad is not None and all(p.grad is None for p in ps[:7])


@pytest.mark.parametrize("per_block", [False, True])
def test_reference_loop_body_verbatim(per_block):
    """train_unconditional.py:120-144 on the drop-in modules reproduces the reference's five Adam steps - on the fused route
    (HintFlow's default) and walking the modules one by one as FrEIA's ReversibleGraphNet does"""
    case = CHAIN_CASES[1]
    c, nodes, shapes, params, perms, xs, g = load_chain_case(case)
    model = build_flow(case, params, perms)
    model.fuse_chain = not per_block
    params_trainable = list(filter(lambda p: p.requires_grad, model.parameters()))
    optim = torch.optim.Adam(params_trainable, lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
    history = []
    for x_np in xs:
        optim.zero_grad()
        x = torch.from_numpy(x_np).to(DEV)
        z = model(x)
        log_jacobian = model.log_jacobian(x, run_forward=False)
        batch_losses = [0.5 * torch.sum(z ** 2, dim=1).mean(), -log_jacobian.mean()]
        loss_total = sum(batch_losses)
        history.append([l.item() for l in batch_losses])
        loss_total.backward()
        for p in params_trainable:
            p.grad.data.clamp_(-5.00, 5.00)
        optim.step()
    np.testing.assert_allclose(np.array(history), g["losses"], rtol=1e-4, atol=1e-5)
    for bi, blk in enumerate(model.blocks):
        for k, v in blk.state_dict().items():
            assert rel_err(v.cpu().numpy(), g[f"final:{bi}:{k}"]) < 1e-3, (bi, k)
            check_update(params[bi][k], v.cpu().numpy(), g[f"final:{bi}:{k}"], (bi, k))


def test_pack_cache_is_opt_in_and_sees_optimizer_updates():
    flow = make_flow()
    x = torch.randn(100, 6, device=DEV)
    blk = flow.blocks[0]
    prev = H.set_pack_cache(True)
    try:
        with torch.no_grad():
            (z0,) = blk([x])
            (z1,) = blk([x])                           # second call: nothing moved, the re-pack is skipped
            assert torch.equal(z0, z1)
            for p in blk.parameters():                 # an in-place update with a version bump (what optimizers do)
                p.mul_(1.5)
            (z2,) = blk([x])
            for p in blk.parameters():                 # rebinding (train_unconditional.py:165-167)
                p.data = p.data * 0.5
            (z3,) = blk([x])
        H.set_pack_cache(False)
        with torch.no_grad():
            (z3b,) = blk([x])
            for p in blk.parameters():
                p.data = p.data * 2.0
            (z2b,) = blk([x])
        assert not torch.equal(z0, z2)
        assert torch.equal(z3, z3b) and torch.equal(z2, z2b)
    finally:
        H.set_pack_cache(prev)


def test_trainer_steps_then_module_forward_sees_the_new_weights():
    """FlowTrainer's kernels update the arena in place (no version bump): a module forward behind it must re-pack"""
    flow = make_flow()
    x = torch.randn(512, 6, device=DEV)
    prev = H.set_pack_cache(True)
    try:
        tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False, lr=1e-2)
        with torch.no_grad():
            flow.fuse_chain = False
            z0 = flow(x).clone()
            for _ in range(3):
                tr.step(x)
            z1 = flow(x).clone()
            flow.fuse_chain = True
            z2 = flow(x)
        assert not torch.allclose(z0, z1)
        np.testing.assert_allclose(z1.cpu().numpy(), z2.cpu().numpy(), rtol=1e-5, atol=1e-5)
    finally:
        H.set_pack_cache(prev)


@pytest.mark.parametrize("fuse", [True, False])
def test_more_live_forwards_than_the_pool_keeps(fuse):
    """four forwards alive at once (a pool keeps two buffers / chain instances; the others are plain allocations released when
    their autograd nodes die), backward in reverse order, then the same again: every gradient as from a lone forward"""
    flow = make_flow(d=6, nb=2, widths=(32, 16))
    flow.fuse_chain = fuse
    ps = list(flow.parameters())
    xs = [torch.randn(200, 6, device=DEV) for _ in range(4)]
    want = [grads_of(flow, x, fuse=fuse)[3] for x in xs]
    flow.fuse_chain = fuse
    for _ in range(2):
        losses = [loss_of(flow, x) for x in xs]
        for k in reversed(range(4)):
            for p in ps:
                p.grad = None
            losses[k].backward()
            for p, g in zip(ps, want[k]):
                assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < 1e-6
        del losses
