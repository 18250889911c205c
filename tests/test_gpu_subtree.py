"""Subtree groups of the general kernels (hint_amd/csrc/hint_sub.hpp: the deepest lean levels of a wide tree, one subtree per
wavefront) through the C ABI against the float32 oracle: forward, inverse, log-det and every gradient tensor, on trees that
exercise what the planner has to get right - cfg 5's tree (hint.py:25-54 on d = 43), both plan variants (8 wavefronts and, above
4096 rows, 4 with two subtrees each), three subtree levels, levels some wavefronts have no nodes in, lanes no subtree covers."""
import numpy as np
import pytest
import torch

import hint_amd
from oracle import hint_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [
    (43, [67, 33, 16, 8], 37, 2),               # cfg 5 (configs/uci_data/miniboone_hint_10.py), ragged batch
    (43, [67, 33, 16, 8], 4100, 2),             # ... on the 4-wavefront variant: two subtrees per wavefront
    (20, [40, 16, 8], 16, 1),
    (64, [16, 16, 16, 16], 70, 0),             # three subtree levels, every wavefront busy in all of them
    (26, [16, 16, 8, 8, 8], 16, 1),             # wavefronts without nodes in the deepest level
    (100, [48, 24, 20, 12, 8, 8], 33, 1),         # 16 subtrees on 8 wavefronts; the deepest level only in every fourth
    (100, [16, 16, 16, 16, 8, 8, 8], 16, 1),
]


@pytest.mark.parametrize("d,widths,B,min_sub", SHAPES, ids=lambda v: str(v).replace(" ", ""))
def test_subtree_groups_vs_oracle(d, widths, B, min_sub):
    nodes = orc.build_nodes(d, [], widths)
    P = orc.init_params(nodes, seed=1, scale=None)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(5))
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=list(widths))
    blk.load_state_dict({k: v.clone() for k, v in P.items()})
    blk = blk.to(DEV)
    info = (__import__("ctypes").c_int32 * 8)()
    eng = blk.tree.engine(torch.device(DEV))
    assert eng.lib.hint_plan_describe(eng.plan, B, info) == 0 and info[0] == 0      # (the general kernels, not the wave-local ones)
    print(f"d={d} widths={widths} B={B}: {info[4]} subtree groups, {info[2]} wavefronts")
    assert info[4] >= min_sub

    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xg = x.clone().requires_grad_(True)
    zo, Jo = orc.block_apply(nodes, Pg, xg, [], rev=False)
    gz = torch.randn(B, d, generator=torch.Generator().manual_seed(6))
    gJ = torch.randn(B, generator=torch.Generator().manual_seed(7))
    ((zo * gz).sum() + (Jo * gJ).sum()).backward()

    xd = x.to(DEV).requires_grad_(True)
    (z,) = blk([xd])
    J = blk.jacobian(None)
    scale = max(1.0, float(zo.abs().max()))
    # (the tolerance of the block tests, tests/test_gpu_parity.py: 1e-5 of the magnitude; default-initialised d = 43 blocks expand
    #  their inputs to |z| ~ 30, and a 4100-row batch holds rows whose error is a few ulp of that)
    assert float((z.detach().cpu() - zo.detach()).abs().max()) < 1e-5 * scale
    assert float((J.detach().cpu() - Jo.detach()).abs().max()) < 1e-5 * max(1.0, float(Jo.abs().max()))
    ((z * gz.to(DEV)).sum() + (J * gJ.to(DEV)).sum()).backward()
    assert float((xd.grad.cpu() - xg.grad).abs().max()) < 1e-4 * float(xg.grad.abs().max())
    sd = dict(blk.named_parameters())
    for k, v in Pg.items():
        err = float((sd[k].grad.cpu() - v.grad).abs().max())
        assert err < 1e-4 * max(1e-6, float(v.grad.abs().max())), (k, err)

    with torch.no_grad():                         # the sampling direction undoes the forward; J_rev = -J_fwd
        (xr,) = blk([z.detach()], rev=True)
        Jr = blk.jacobian(None, rev=True)
    assert float((xr.cpu() - x).abs().max()) < 1e-4 * scale
    assert float((J.detach() + Jr).abs().max()) < 1e-4 * max(1.0, float(Jo.abs().max()))


def test_subtree_groups_equal_general_groups(monkeypatch):
    """the same block planned with and without subtree groups (HINT_SUB=0 at plan time): same function, same gradients"""
    d, widths, B = 43, [67, 33, 16, 8], 100
    nodes = orc.build_nodes(d, [], widths)
    P = orc.init_params(nodes, seed=3, scale=None)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(8)).to(DEV)
    outs = []
    from hint_amd import _lib
    lib = _lib.load()
    for sub in ("1", "0"):
        monkeypatch.setenv("HINT_SUB", sub)
        lib.hint_debug_reload_knobs()             # (the library reads its environment once)
        blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=list(widths))
        blk.load_state_dict({k: v.clone() for k, v in P.items()})
        blk = blk.to(DEV)
        xd = x.clone().requires_grad_(True)
        (z,) = blk([xd])
        J = blk.jacobian(None)
        (0.5 * (z ** 2).sum(1).mean() - J.mean()).backward()
        outs.append((z.detach().cpu(), J.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in blk.named_parameters()}))
    monkeypatch.delenv("HINT_SUB")
    lib.hint_debug_reload_knobs()
    (z1, J1, g1, p1), (z0, J0, g0, p0) = outs
    np.testing.assert_allclose(z1.numpy(), z0.numpy(), rtol=1e-5, atol=1e-5 * float(z0.abs().max()))
    np.testing.assert_allclose(J1.numpy(), J0.numpy(), rtol=1e-5, atol=1e-5 * float(J0.abs().max()))
    np.testing.assert_allclose(g1.numpy(), g0.numpy(), rtol=1e-4, atol=1e-5 * float(g0.abs().max()))
    for k in p0:
        np.testing.assert_allclose(p1[k].numpy(), p0[k].numpy(), rtol=1e-4, atol=2e-5 * max(1e-6, float(p0[k].abs().max())), err_msg=k)


@pytest.mark.parametrize("d,widths,B", [(43, [67, 33, 16, 8], 333), (100, [224, 112, 56], 200), (22, [40, 24], 4112)])
def test_lean_wide_groups_equal_stored_operands(monkeypatch, d, widths, B):
    """lean-wide groups (round 6: thin layers with 5 .. HINT_LEANW_MAX inputs / outputs - the forward stores no a1, the backward no
    g2, part B rebuilds both with chained K = 4 MFMAs) against the same block with every operand kept in HBM (HINT_LEANW=0): the
    forward is bit-identical (only stores are skipped), every gradient agrees to rounding; at the default width limit (12) and
    at the widest the kernels take (28: MINIBOONE's root with 21 inputs / 22 outputs, the d = 100 tree's 25-wide level)"""
    from hint_amd import _lib
    lib = _lib.load()
    torch.manual_seed(4)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(9)).to(DEV)
    ref_flow = hint_amd.HintFlow(d, 2, widths)
    state = {k: v.clone() + 0.05 * torch.randn_like(v) for k, v in ref_flow.state_dict().items()}
    outs = {}
    try:
        for name, env in (("off", {"HINT_LEANW": "0"}), ("default", {}), ("max", {"HINT_LEANW_MAX": "28"})):
            for k in ("HINT_LEANW", "HINT_LEANW_MAX"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            lib.hint_debug_reload_knobs()
            flow = hint_amd.HintFlow(d, 2, widths)
            flow.load_state_dict(state)
            flow = flow.to(DEV)
            tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
            tr._check_arenas()
            tr.G.zero_()
            tr._fwd_bwd(x, None)
            torch.cuda.synchronize()
            outs[name] = (tr.loss_acc.clone(), tr.G.clone())
            del tr, flow
    finally:
        for k in ("HINT_LEANW", "HINT_LEANW_MAX"):
            monkeypatch.delenv(k, raising=False)
        lib.hint_debug_reload_knobs()
    gref = outs["off"][1].double()
    for name in ("default", "max"):
        if B <= 1024:           # one workgroup per loss slot: the forward's loss sums bit for bit (beyond, the slots' atomic adds come in any order)
            assert torch.equal(outs[name][0], outs["off"][0]), name
        else:
            assert torch.allclose(outs[name][0].sum(0), outs["off"][0].sum(0), rtol=1e-6), name
        g = outs[name][1].double()
        assert float((g - gref).norm() / gref.norm()) < 2e-6, (name, float((g - gref).norm() / gref.norm()))
        assert float((g - gref).abs().max()) < 1e-5 * float(gref.abs().max()), name
