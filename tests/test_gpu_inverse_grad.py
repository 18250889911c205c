"""Gradients THROUGH the inverse direction on the GPU: `block([z], c, rev=True)` under autograd, as the reference's
module allows (hint.py:82-88 are plain differentiable torch ops; its own loops only sample under no_grad).
Checked against (1) golden vectors of the real reference (tests/golden/revgrad_*.npz, make_golden.py::gen_block_rev),
(2) the CPU oracle's autograd on seeded inputs at several batch sizes, (3) the identity that ties the two directions
together: with x = block^-1(z), the gradients of f(block(x)) and of the same f through the round trip agree.

Tolerances: 1e-4 relative to each tensor's max-abs, like the forward direction's gradients (test_gpu_parity.py); the
inverse divides by e(s), so ill-conditioned blocks (the big_s fixture) are left to the forward tests."""
import numpy as np
import pytest
import torch

from oracle import hint_oracle as orc
from test_gpu_parity import DEV, make_block
from util import BLOCK_CASES, REV_GRAD_CASES, case_perms, load_block_case, load_revgrad, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [c for c in BLOCK_CASES if c["name"] in REV_GRAD_CASES], ids=lambda c: c["name"])
def test_inverse_grads_vs_reference_golden(case):
    c, nodes, shapes, params, z_np, conds_np, g = load_block_case(case)
    r = load_revgrad(c, g)
    blk = make_block(c, params, case_perms(g))
    z = torch.from_numpy(z_np).to(DEV).requires_grad_(True)
    conds = [torch.from_numpy(a).to(DEV).requires_grad_(True) for a in conds_np]
    (x,) = blk([z], c=conds, rev=True)
    J = blk.jacobian(None, rev=True)
    assert rel_err(x.detach().cpu().numpy(), r["x_inv"]) < 1e-4
    L = (0.5 * torch.sum(x ** 2, dim=1) - J).mean()
    assert abs(L.item() - float(r["L"])) <= 1e-4 * max(1.0, abs(float(r["L"])))
    L.backward()
    assert rel_err(z.grad.cpu().numpy(), r["gz"]) < 1e-4
    for i, cc in enumerate(conds):
        assert rel_err(cc.grad.cpu().numpy(), r[f"gc{i}"]) < 1e-4
    named = dict(blk.named_parameters())
    for k in shapes:
        assert rel_err(named[k].grad.cpu().numpy(), r["g:" + k]) < 1e-4, k


@pytest.mark.parametrize("d,widths,dc,B", [
    (6, [140, 70, 35, 17], 0, 4096),      # BASELINE config 2 (one block): wave-local level plans
    (8, [128, 64, 32, 16], 0, 1000),
    (43, [67, 33, 16, 8], 0, 515),
    (100, [224, 112, 56], 4, 77),
    (6, [200, 100, 50, 25], 0, 1),
    (9, [19, 11, 3], 2, 4113),            # more than one row tile per workgroup, ragged
])
def test_inverse_grads_vs_oracle_seeded(d, widths, dc, B):
    dims_c = [(dc,)] if dc else []
    nodes = orc.build_nodes(d, dims_c, widths)
    P = orc.init_params(nodes, seed=13, scale=None)
    gen = torch.Generator().manual_seed(6)
    z = torch.randn(B, d, generator=gen)
    cond = [torch.randn(B, dc, generator=gen)] if dc else []
    w = torch.randn(B, generator=gen)                   # a J weight per row: g_J is not constant
    c = dict(d=d, dims_c=dims_c, c_internal=widths, clamp=4.0, max_splits=-1, min_split_size=2)
    blk = make_block(c, {k: v.numpy() for k, v in P.items()})

    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    zo = z.clone().requires_grad_(True)
    co = [t.clone().requires_grad_(True) for t in cond]
    xo, Jo = orc.block_apply(nodes, Po, zo, co, rev=True)
    (0.5 * torch.sum(xo ** 2, dim=1) - w * Jo).mean().backward()

    zg = z.to(DEV).requires_grad_(True)
    cg = [t.to(DEV).requires_grad_(True) for t in cond]
    (xg,) = blk([zg], c=cg, rev=True)
    Jg = blk.jacobian(None, rev=True)
    (0.5 * torch.sum(xg ** 2, dim=1) - w.to(DEV) * Jg).mean().backward()
    assert rel_err(xg.detach().cpu().numpy(), xo.detach().numpy()) < 1e-4
    assert rel_err(zg.grad.cpu().numpy(), zo.grad.numpy()) < 1e-4
    for a, b in zip(cg, co):
        assert rel_err(a.grad.cpu().numpy(), b.grad.numpy()) < 1e-4
    named = dict(blk.named_parameters())
    for k in Po:
        assert rel_err(named[k].grad.cpu().numpy(), Po[k].grad.numpy()) < 1e-4, k


def test_inverse_then_forward_is_the_identity_for_gradients():
    """z -> x = block^-1(z) -> z' = block(x): z' = z for every parameter value, so d f(z') / dz = f'(z) and every
    parameter gradient of the round trip vanishes - the two backward passes must cancel each other."""
    d, widths = 8, [64, 32, 16]
    nodes = orc.build_nodes(d, [], widths)
    P = orc.init_params(nodes, seed=3, scale=None)
    c = dict(d=d, dims_c=[], c_internal=widths, clamp=4.0, max_splits=-1, min_split_size=2)
    blk = make_block(c, {k: v.numpy() for k, v in P.items()})
    z = torch.randn(300, d, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
    (x,) = blk([z], rev=True)
    Jr = blk.jacobian(None, rev=True)
    (z2,) = blk([x])
    Jf = blk.jacobian(None)
    wv = torch.linspace(-1, 1, d, device=DEV)
    L = (torch.sin(z2) * wv).sum(dim=1).mean() + (Jr + Jf).mean()          # J_rev + J_fwd = 0 identically
    L.backward()
    want = (torch.cos(z.detach()) * wv) / z.shape[0]
    assert rel_err(z.grad.cpu().numpy(), want.cpu().numpy()) < 1e-4
    gmax = max(float(p.grad.abs().max()) for p in blk.parameters())
    # (scale: what ONE of the two cancelling passes contributes)
    blk.zero_grad()
    (x1,) = blk([z.detach()], rev=True)
    (z3,) = blk([x1.detach()])
    ((torch.sin(z3) * wv).sum(dim=1).mean()).backward()
    one = max(float(p.grad.abs().max()) for p in blk.parameters())
    assert gmax < 2e-4 * max(one, 1e-6), (gmax, one)


def test_flow_inverse_grads_vs_oracle_flow():
    """the flow container's rev=True path (block inverses and fixed permutations, last block first) under autograd
    against the oracle flow carrying the same weights and matrices"""
    import hint_amd
    d, widths, nb, B = 6, [24, 12], 3, 200
    torch.manual_seed(0)
    flow = hint_amd.HintFlow(d, nb, widths).to(DEV)
    of = orc.OracleFlow(d, nb, widths)
    of.params = [{k: v.detach().cpu().clone().requires_grad_(True) for k, v in flow.blocks[i].state_dict().items()
                  if k in of.params[i]} for i in range(nb)]
    assert all(list(P.keys()) == list(orc.param_shapes(of.nodes).keys()) for P in of.params)
    of.perms = [flow.perms[i].W.detach().cpu().clone() if flow.has_perm(i) else None for i in range(nb)]
    z = torch.randn(B, d, generator=torch.Generator().manual_seed(2))
    zo = z.clone().requires_grad_(True)
    xo, Jo = of.inverse(zo)
    (0.5 * torch.sum(xo ** 2, dim=1) - Jo).mean().backward()
    zg = z.to(DEV).requires_grad_(True)
    xg = flow(zg, rev=True)
    (0.5 * torch.sum(xg ** 2, dim=1) - flow.log_jacobian(run_forward=False)).mean().backward()
    assert rel_err(xg.detach().cpu().numpy(), xo.detach().numpy()) < 1e-4
    assert rel_err(zg.grad.cpu().numpy(), zo.grad.numpy()) < 1e-4
    for i in range(nb):
        named = dict(flow.blocks[i].named_parameters())
        for k, v in of.params[i].items():
            assert rel_err(named[k].grad.cpu().numpy(), v.grad.numpy()) < 1e-4, (i, k)


@pytest.mark.parametrize("D,dc,h,B", [(5, 2, 16, 77), (100, 4, 224, 300), (1, 3, 8, 16)])
def test_external_coupling_inverse_grads(D, dc, h, B):
    """the conditional model's couplings (conditional_hint_4_full.py:76-89: k = 0, every lane transformed given y) through
    the inverse under autograd, against the oracle's definition of the same node"""
    import hint_amd
    from test_gpu_conditional import oracle_nodes
    torch.manual_seed(1)
    mod = hint_amd.ExternalAffineCoupling([(D,)], dims_c=[(dc,)], F_args={"internal_size": h}).to(DEV)
    z = torch.randn(B, D); c = torch.randn(B, dc)
    zg = z.to(DEV).requires_grad_(True); cg = c.to(DEV).requires_grad_(True)
    (x,) = mod([zg], c=[cg], rev=True); J = mod.jacobian(None, rev=True)
    zo = z.clone().requires_grad_(True); co = c.clone().requires_grad_(True)
    Po = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    xo, Jo = orc.block_apply(oracle_nodes(mod.tree, dc), Po, zo, [co], rev=True, clamp=mod.tree.clamp)
    (0.5 * (x ** 2).sum(1) - J).mean().backward()
    (0.5 * (xo ** 2).sum(1) - Jo).mean().backward()
    assert rel_err(x.detach().cpu().numpy(), xo.detach().numpy()) < 1e-4
    assert rel_err(zg.grad.cpu().numpy(), zo.grad.numpy()) < 1e-4
    assert rel_err(cg.grad.cpu().numpy(), co.grad.numpy()) < 1e-4
    for k, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), Po[k].grad.numpy()) < 2e-4, k


def test_inverse_backward_entry_point_contract():
    """hint_block_inverse_backward directly: accumulate adds to g_params, NULL upstream gradients mean zeros, a short
    workspace and a missing condition are refused, an empty batch clears the gradient"""
    import ctypes as C
    from hint_amd import _lib
    d, dc, widths, B = 9, 2, [19, 11, 3], 50
    c = dict(d=d, dims_c=[(dc,)], c_internal=widths, clamp=4.0, max_splits=-1, min_split_size=2)
    blk = make_block(c)
    z = torch.randn(B, d, device=DEV); cond = torch.randn(B, dc, device=DEV)
    with torch.no_grad():
        (x,) = blk([z], c=[cond], rev=True)
    eng = blk.tree.engine(torch.device(DEV))
    lib = eng.lib
    nbytes = lib.hint_plan_inverse_workspace_bytes(eng.plan, B)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    gx = torch.randn(B, d, device=DEV); gJ = torch.randn(B, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def call(gx, gJ, G, acc, nb=nbytes, cptr=cond.data_ptr(), rows=B):
        gz = torch.empty(B, d, device=DEV); gc = torch.empty(B, dc, device=DEV)
        st = lib.hint_block_inverse_backward(eng.plan, eng.arena.data_ptr(), x.data_ptr(), cptr,
                                             gx.data_ptr() if gx is not None else None, gJ.data_ptr() if gJ is not None else None,
                                             gz.data_ptr(), gc.data_ptr(), G.data_ptr(), acc, ws.data_ptr(), nb, None, rows, stream)
        torch.cuda.synchronize()
        return st, gz, gc
    G1 = torch.full((eng.total,), 7.0, device=DEV)
    st, gz1, gc1 = call(gx, gJ, G1, 0)
    assert st == 0
    G2 = G1.clone()
    st, gz2, gc2 = call(gx, gJ, G2, 1)
    assert st == 0 and torch.equal(gz1, gz2) and torch.equal(gc1, gc2)
    assert torch.allclose(G2, 2 * G1, rtol=1e-6, atol=1e-7)
    # linear in the upstream pair: (g_x, 0) + (0, g_J) = (g_x, g_J)
    Ga, Gb = torch.empty_like(G1), torch.empty_like(G1)
    _, gza, _ = call(gx, None, Ga, 0)
    _, gzb, _ = call(None, gJ, Gb, 0)
    assert rel_err((gza + gzb).cpu().numpy(), gz1.cpu().numpy()) < 1e-5
    assert rel_err((Ga + Gb).cpu().numpy(), G1.cpu().numpy()) < 1e-5
    assert call(gx, gJ, G2, 0, nb=nbytes - 16)[0] != 0 and b"workspace" in lib.hint_last_error()
    assert call(gx, gJ, G2, 0, cptr=None)[0] != 0
    st, _, _ = call(gx, gJ, G2, 0, rows=0)
    assert st == 0 and float(G2.abs().max()) == 0.0
