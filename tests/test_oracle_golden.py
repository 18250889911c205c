"""Pin the CPU oracle (oracle/hint_oracle.py) to golden vectors produced by the real
reference hint.py (tests/golden/make_golden.py).  Tolerances: the oracle and the reference
run the same ATen ops, but the flat schedule sums J in a different order and the GPU box's
CPU may take different BLAS paths, so compare at rtol 1e-5 / atol 1e-5 (observed ≤ 2e-6)."""
import math

import numpy as np
import pytest
import torch

from oracle import hint_oracle as orc
from util import (BLOCK_CASES, CHAIN_CASES, REV_GRAD_CASES, case_perms, load_block_case, load_chain_case, load_revgrad, rel_err,
                  to_torch)

TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", BLOCK_CASES, ids=lambda c: c["name"])
def test_block_forward_inverse_grads(case):
    c, nodes, shapes, params, x_np, conds_np, g = load_block_case(case)
    P = to_torch(params)
    for p in P.values():
        p.requires_grad_(True)
    x = torch.from_numpy(x_np).requires_grad_(True)
    conds = [torch.from_numpy(a).requires_grad_(True) for a in conds_np]
    perms = {k: torch.from_numpy(v) for k, v in case_perms(g).items()}
    assert bool(perms) == c["reshuffle"]
    z, J = orc.block_apply(nodes, P, x, conds, rev=False, clamp=c["clamp"], perms=perms)
    np.testing.assert_allclose(z.detach().numpy(), g["z"], **TOL)
    np.testing.assert_allclose(J.detach().numpy(), g["J"], **TOL)
    L = (0.5 * torch.sum(z ** 2, dim=1) - J).mean()
    assert abs(L.item() - float(g["L"])) <= 1e-5 * max(1.0, abs(float(g["L"])))
    L.backward()
    assert rel_err(x.grad.numpy(), g["gx"]) < 1e-4
    for i, cc in enumerate(conds):
        assert rel_err(cc.grad.numpy(), g[f"gc{i}"]) < 1e-4
    for k in shapes:
        assert rel_err(P[k].grad.numpy(), g["g:" + k]) < 1e-4, k
    with torch.no_grad():
        xr, Jr = orc.block_apply(nodes, P, z.detach(), [cc.detach() for cc in conds], rev=True, clamp=c["clamp"], perms=perms)
        xi, Ji = orc.block_apply(nodes, P, x.detach(), [cc.detach() for cc in conds], rev=True, clamp=c["clamp"], perms=perms)
    scale = max(1.0, float(np.abs(g["x_rec"]).max()))
    np.testing.assert_allclose(xr.numpy(), g["x_rec"], rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(Jr.numpy(), g["J_rev"], **TOL)
    assert rel_err(xi.numpy(), g["x_inv"]) < 1e-4
    np.testing.assert_allclose(Ji.numpy(), g["J_inv"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", [c for c in BLOCK_CASES if c["name"] in REV_GRAD_CASES], ids=lambda c: c["name"])
def test_block_grads_through_inverse(case):
    """rev=True under autograd (hint.py:82-88): d/dz, d/dc, d/dW of mean(0.5|x|^2 - J), (x, J) = block(z, rev=True)"""
    c, nodes, shapes, params, z_np, conds_np, g = load_block_case(case)
    r = load_revgrad(c, g)
    P = to_torch(params)
    for p in P.values():
        p.requires_grad_(True)
    z = torch.from_numpy(z_np).requires_grad_(True)
    conds = [torch.from_numpy(a).requires_grad_(True) for a in conds_np]
    perms = {k: torch.from_numpy(v) for k, v in case_perms(g).items()}
    x, J = orc.block_apply(nodes, P, z, conds, rev=True, clamp=c["clamp"], perms=perms)
    assert rel_err(x.detach().numpy(), r["x_inv"]) < 1e-4
    L = (0.5 * torch.sum(x ** 2, dim=1) - J).mean()
    assert abs(L.item() - float(r["L"])) <= 1e-4 * max(1.0, abs(float(r["L"])))
    L.backward()
    assert rel_err(z.grad.numpy(), r["gz"]) < 1e-4
    for i, cc in enumerate(conds):
        assert rel_err(cc.grad.numpy(), r[f"gc{i}"]) < 1e-4
    for k in shapes:
        assert rel_err(P[k].grad.numpy(), r["g:" + k]) < 1e-4, k


@pytest.mark.parametrize("case", BLOCK_CASES[:6], ids=lambda c: c["name"])
def test_reference_selfconsistency_properties(case):
    """Properties the reference itself satisfies: round trip and J_fwd + J_rev = 0."""
    c, nodes, shapes, params, x_np, conds_np, g = load_block_case(case)
    assert np.max(np.abs(g["x_rec"] - x_np)) < 2e-4 * max(1.0, np.abs(x_np).max())
    assert np.max(np.abs(g["J"] + g["J_rev"])) < 1e-4


def test_logdet_matches_autograd_jacobian():
    """J == log|det dz/dx| and the Jacobian is lower-triangular in lane order."""
    nodes = orc.build_nodes(6, (), [12, 6])
    P = orc.init_params(nodes, seed=3, scale=None, dtype=torch.float64)
    x = torch.randn(6, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    f = lambda v: orc.block_apply(nodes, P, v[None], rev=False)[0][0]
    Jm = torch.autograd.functional.jacobian(f, x)
    _, J = orc.block_apply(nodes, P, x[None], rev=False)
    assert abs(torch.linalg.slogdet(Jm)[1].item() - J.item()) < 1e-10
    assert torch.triu(Jm, diagonal=1).abs().max().item() == 0.0


@pytest.mark.parametrize("case", CHAIN_CASES, ids=lambda c: c["name"])
def test_chain(case):
    c, nodes, shapes, params, perms, xs, g = load_chain_case(case)
    flow = orc.OracleFlow(case["d"], case["n_blocks"], case["c_internal"])
    flow.params = [to_torch(P) for P in params]
    flow.perms = [None if p is None else torch.from_numpy(p) for p in perms]
    if case["steps"] == 0:
        with torch.no_grad():
            z, J = flow.forward(torch.from_numpy(xs[0]))
        np.testing.assert_allclose(z.numpy(), g["z"], **TOL)
        np.testing.assert_allclose(J.numpy(), g["J"], **TOL)
        assert abs(flow.nll(z, J) - float(g["nll"])) < 1e-5 * abs(float(g["nll"]))
    else:
        flow.make_optimizer()
        losses = [flow.train_step(torch.from_numpy(x)) for x in xs]
        np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-5, atol=1e-5)
        for bi, P in enumerate(flow.params):
            for k, v in P.items():
                np.testing.assert_allclose(v.detach().numpy(), g[f"final:{bi}:{k}"], rtol=1e-4, atol=2e-6)
