"""GPU tests of the chained flow and the fast training step against the reference-generated
chain fixtures (tests/golden/chain_*.npz)."""
import numpy as np
import pytest
import torch

import hint_amd
from util import CHAIN_CASES, load_chain_case, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_flow(case, params, perms):
    flow = hint_amd.HintFlow(case["d"], case["n_blocks"], case["c_internal"])
    for i, blk in enumerate(flow.blocks):
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in params[i].items()})
        if perms[i] is not None:
            flow.perms[i].W.copy_(torch.from_numpy(np.ascontiguousarray(perms[i])))
    return flow.to(DEV)


def test_chain_forward_nll_matches_reference():
    case = CHAIN_CASES[0]
    c, nodes, shapes, params, perms, xs, g = load_chain_case(case)
    flow = build_flow(case, params, perms)
    x = torch.from_numpy(xs[0]).to(DEV)
    with torch.no_grad():
        z = flow(x)
        J = flow.log_jacobian(run_forward=False)
        xr = flow(z, rev=True)
    np.testing.assert_allclose(z.cpu().numpy(), g["z"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(J.cpu().numpy(), g["J"], rtol=1e-5, atol=1e-5)
    nll = float(0.5 * (z ** 2).sum(1).mean() - J.mean()) + 0.5 * case["d"] * np.log(2 * np.pi)
    assert abs(nll - float(g["nll"])) < 1e-4 * abs(float(g["nll"]))      # north_star: 1e-4 relative
    assert (xr - x).abs().max().item() < 1e-4


def check_update(initial, final, final_ref, what):
    """the MOVEMENT of a tensor over the fixture's steps against the reference's: the weights themselves barely move (five
    Adam steps of 3e-4), so a tolerance on the weights alone would not see a gradient of the wrong sign in a small tensor -
    the update vector does (a flipped sign turns it round: relative deviation 2)"""
    upd = np.asarray(final, dtype=np.float64) - np.asarray(initial, dtype=np.float64)
    ref = np.asarray(final_ref, dtype=np.float64) - np.asarray(initial, dtype=np.float64)
    dev = np.linalg.norm(upd - ref) / max(np.linalg.norm(ref), 1e-30)
    assert dev < 5e-2, (what, dev)


@pytest.mark.parametrize("use_graph", [False, True])
def test_trainer_reproduces_reference_adam_steps(use_graph):
    """K=5 steps of train_unconditional.py:120-144 (noise off) from fixed weights: per-step
    loss pair and final weights as the reference produced them."""
    case = CHAIN_CASES[1]
    c, nodes, shapes, params, perms, xs, g = load_chain_case(case)
    flow = build_flow(case, params, perms)
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=use_graph)
    losses = []
    for x_np in xs:
        l0, l1 = tr.step(torch.from_numpy(x_np).to(DEV))
        losses.append([float(l0), float(l1)])
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-4, atol=1e-5)
    for bi, blk in enumerate(flow.blocks):
        for k, v in blk.state_dict().items():
            assert rel_err(v.cpu().numpy(), g[f"final:{bi}:{k}"]) < 1e-3, (bi, k)
            np.testing.assert_allclose(v.cpu().numpy(), g[f"final:{bi}:{k}"], rtol=2e-3, atol=3e-5)
            check_update(params[bi][k], v.cpu().numpy(), g[f"final:{bi}:{k}"], (bi, k))


@pytest.mark.parametrize("reshuffle", [False, True])
def test_autograd_path_equals_fast_path(reshuffle):
    """the nn.Module/autograd route (drop-in for the reference loop) and the FlowTrainer route
    give the same gradients"""
    torch.manual_seed(0)
    flow = hint_amd.HintFlow(8, 3, [64, 32, 16], reshuffle=reshuffle).to(DEV)
    x = torch.randn(333, 8, device=DEV)
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    z = flow(x)
    J = flow.log_jacobian(run_forward=False)
    (0.5 * (z ** 2).sum(1).mean() - J.mean()).backward()
    ref = {n: p.grad.clone() for n, p in flow.named_parameters()}
    tr._check_arenas()
    tr.G.zero_()
    tr._fwd_bwd(x, None)
    for (a, b), eng, blk_i in zip(tr.slices, tr.engines, range(3)):
        for p, g in zip(eng.params, eng.split_flat(tr.G[a:b])):
            name = [n for n, q in flow.named_parameters() if q is p][0]
            assert rel_err(g.cpu().numpy(), ref[name].cpu().numpy()) < 1e-5, name


def test_adam_kernel_matches_torch():
    from hint_amd import _lib
    lib = _lib.load()
    n = 10007
    torch.manual_seed(1)
    p = torch.randn(n + 1, device=DEV)[:n].clone()
    g = 20 * torch.randn(n, device=DEV)
    p_ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=3e-4, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        p_ref.grad = (g * 0.5).clamp(-5, 5)
        opt.step()
        st = lib.hint_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, step, 3e-4, 0.9, 0.95,
                                1e-4, 1.86e-5, 0.5, 5.0, 0, torch.cuda.current_stream().cuda_stream)
        assert st == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_reference_training_loop_body_on_module_path():
    """The statements of train_unconditional.py:120-144 executed verbatim on the drop-in modules
    (nn.Module + autograd route, torch.optim.Adam, per-parameter clamp) reproduce the losses and
    weights the reference produced from the same start."""
    case = CHAIN_CASES[1]
    c, nodes, shapes, params, perms, xs, g = load_chain_case(case)
    model = build_flow(case, params, perms)
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


@pytest.mark.parametrize("d,dc,widths,perm_first,n_blocks,B,reshuffle", [
    (6, 0, [32, 16], False, 3, 333, False),
    (8, 3, [64, 32, 16], True, 4, 1000, False),
    (2, 0, [16], False, 2, 50, False),
    (3, 2, [24, 8], True, 1, 17, False),
    (21, 0, [48, 40, 24, 16], False, 2, 5000, False),
    (6, 0, [448, 64], False, 2, 300, False),               # split (h > 384) root
    (7, 0, [24, 12, 6], False, 3, 200, True),              # node permutations folded into the chain's matrices
    (6, 0, [16, 8], False, 2, 40007, False),               # more row tiles than workgroups (persistent loop), ragged
    (70, 0, [24, 12], True, 2, 100, False),                # permutation matrices too big for the LDS table (read from global memory)
])
def test_chain_launch_equals_per_block_launches(d, dc, widths, perm_first, n_blocks, B, reshuffle):
    """hint_chain_forward / hint_chain_backward (one launch for all blocks) against the same
    step issued block by block through hint_block_*_ex"""
    torch.manual_seed(3)
    flow = hint_amd.HintFlow(d, n_blocks, widths, ndim_c=dc, perm_first=perm_first, reshuffle=reshuffle).to(DEV)
    for p in flow.parameters():                      # away from the near-identity init
        p.data.add_(0.05 * torch.randn_like(p))
    x = torch.randn(B, d, device=DEV)
    c = torch.randn(B, dc, device=DEV) if dc > 0 else None
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    assert tr._chainable
    out = {}
    for mode in (False, True):
        tr._chainable = mode
        tr._check_arenas()
        tr.G.zero_()
        tr._fwd_bwd(x, c)
        torch.cuda.synchronize()
        out[mode] = (tr.G.clone(), tr.loss_acc.sum(0).clone())
    assert rel_err(out[True][1].cpu().numpy(), out[False][1].cpu().numpy()) < 1e-6
    assert rel_err(out[True][0].cpu().numpy(), out[False][0].cpu().numpy()) < 1e-5


def test_in_kernel_noise_is_standard_normal_and_consistent():
    """hint_chain_forward_noisy: x_noisy - x is noise * N(0,1) (train_unconditional.py:121), changes
    with the step counter, repeats for the same (seed, step), and z, J are the plain forward of
    x_noisy"""
    from hint_amd import _lib
    torch.manual_seed(5)
    d, B, noise = 6, 8192, 0.25
    flow = hint_amd.HintFlow(d, 3, [32, 16]).to(DEV)
    tr = hint_amd.FlowTrainer(flow, noise=noise, use_graph=False, seed=1234)
    lib, chain = tr.lib, tr._chain_for(B)
    x = torch.randn(B, d, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def run(step):
        tr.rng_state[1] = step
        z = torch.empty_like(x); J = torch.empty(B, device=DEV); xn = torch.empty_like(x)
        _lib.check(lib.hint_chain_forward_noisy(chain, x.data_ptr(), None, z.data_ptr(), J.data_ptr(), None, None,
                                                noise, tr.rng_state.data_ptr(), xn.data_ptr(), st), "noisy")
        torch.cuda.synchronize()
        return z, J, xn

    z1, J1, xn1 = run(1)
    z1b, _, xn1b = run(1)
    z2, _, xn2 = run(2)
    assert torch.equal(xn1, xn1b) and torch.equal(z1, z1b)
    e1, e2 = ((xn1 - x) / noise).double().flatten(), ((xn2 - x) / noise).double().flatten()
    n = e1.numel()
    for e in (e1, e2):
        assert abs(e.mean().item()) < 5.0 / np.sqrt(n)
        assert abs(e.var().item() - 1.0) < 0.03
        assert abs((e ** 4).mean().item() - 3.0) < 0.2            # kurtosis of a Gaussian
        assert abs((e ** 3).mean().item()) < 0.1
    assert abs(torch.dot(e1, e2).item() / n) < 5.0 / np.sqrt(n)   # different steps are uncorrelated
    # neighbouring elements / rows are uncorrelated too
    E = ((xn1 - x) / noise).double()
    assert abs((E[:, 0] * E[:, 1]).mean().item()) < 5.0 / np.sqrt(B)
    assert abs((E[:-1, 0] * E[1:, 0]).mean().item()) < 5.0 / np.sqrt(B)
    zc = torch.empty_like(x); Jc = torch.empty(B, device=DEV)
    _lib.check(lib.hint_chain_forward(chain, xn1.data_ptr(), None, zc.data_ptr(), Jc.data_ptr(), None, None, st), "plain")
    torch.cuda.synchronize()
    assert torch.equal(zc, z1) and torch.equal(Jc, J1)


def test_trainer_step_prologue_clears_losses_and_advances_counter():
    torch.manual_seed(6)
    flow = hint_amd.HintFlow(6, 2, [32, 16]).to(DEV)
    tr = hint_amd.FlowTrainer(flow, noise=0.01, use_graph=True, seed=7)
    x = torch.randn(512, 6, device=DEV)
    vals = []
    for k in range(3):
        l0, l1 = tr.step(x)
        vals.append((float(l0), float(l1)))
        assert int(tr.rng_state[1].item()) >= k + 1
    # the sums are per step (not running totals): the same batch gives nearly the same loss pair
    assert abs(vals[2][0] - vals[0][0]) < 0.2 * abs(vals[0][0]) + 0.1


def test_input_buffers_skip_the_per_step_copy():
    """a batch written into the captured step's own input buffer (FlowTrainer.input_buffers) gives the
    same steps as one passed by value (and copied) every iteration"""
    import copy
    torch.manual_seed(7)
    flow1 = hint_amd.HintFlow(6, 3, [32, 16]).to(DEV)
    for p in flow1.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    flow2 = copy.deepcopy(flow1)
    xs = [torch.randn(300, 6, device=DEV) for _ in range(3)]
    t1 = hint_amd.FlowTrainer(flow1, noise=0.0, use_graph=True)
    t2 = hint_amd.FlowTrainer(flow2, noise=0.0, use_graph=True)
    buf, _ = t2.input_buffers(xs[0])
    for x in xs:
        t1.step(x)
        a = [float(v) for v in t1.last_losses()]
        buf.copy_(x)                                   # the "data pipeline" writes the batch in place
        t2.step(buf)
        b = [float(v) for v in t2.last_losses()]
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), (a, b)        # (float atomics: not bit-reproducible)
    for p1, p2 in zip(flow1.parameters(), flow2.parameters()):
        assert rel_err(p2.detach().cpu().numpy(), p1.detach().cpu().numpy()) < 1e-4


def test_step_many_equals_single_steps():
    """K iterations in one graph replay (FlowTrainer.step_many) take the same steps, with the same
    per-iteration losses, as K step() calls; a learning-rate change between two replays needs no re-capture"""
    import copy
    torch.manual_seed(8)
    flow1 = hint_amd.HintFlow(6, 3, [32, 16]).to(DEV)
    for p in flow1.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    flow2 = copy.deepcopy(flow1)
    K, B = 3, 200
    t1 = hint_amd.FlowTrainer(flow1, noise=0.0, use_graph=True)
    t2 = hint_amd.FlowTrainer(flow2, noise=0.0, use_graph=True)
    for rep in range(2):
        xs = torch.randn(K, B, 6, device=DEV)
        if rep == 1:
            t1.lr = 0.5 * t1.lr; t2.lr = 0.5 * t2.lr
        want = []
        for k in range(K):
            t1.step(xs[k])
            want.append([float(v) for v in t1.last_losses()])
        t2.step_many(xs)
        got = t2.step_losses().cpu().numpy()
        assert np.allclose(got, np.array(want), rtol=1e-5, atol=1e-6), (got, want)
    assert t1.step_count == t2.step_count == 2 * K
    for p1, p2 in zip(flow1.parameters(), flow2.parameters()):
        assert rel_err(p2.detach().cpu().numpy(), p1.detach().cpu().numpy()) < 1e-4
    t2.step(xs[0])                                    # the single-step path still works afterwards
    assert np.isfinite(float(t2.last_losses()[0]))


def test_module_copies_and_pickles_after_first_use(tmp_path):
    """a module that already ran (it owns a plan handle and device arenas) can be deep-copied (EMA / best-checkpoint
    copies mid-training) and saved whole; the copy builds its own engine and computes the same thing"""
    import copy
    torch.manual_seed(0)
    flow = hint_amd.HintFlow(6, 2, [32, 16]).to(DEV)
    x = torch.randn(50, 6, device=DEV)
    with torch.no_grad():
        z = flow(x)
        J = flow.log_jacobian(run_forward=False)
        twin = copy.deepcopy(flow)
        assert twin.blocks[0].tree._engine is None and flow.blocks[0].tree._engine is not None
        assert torch.equal(twin(x), z) and torch.equal(twin.log_jacobian(run_forward=False), J)
        torch.save(flow, tmp_path / "flow.pt")
        back = torch.load(tmp_path / "flow.pt", weights_only=False)
        assert torch.equal(back(x), z)
        # the copies own their weights
        for p in twin.parameters():
            p.mul_(0.5)
        assert torch.equal(flow(x), z) and not torch.equal(twin(x), z)


@pytest.mark.parametrize("d,widths,nb,B", [(6, [140, 70, 35, 17], 3, 1000), (43, [67, 33, 16, 8], 2, 300), (9, [19, 11, 3], 4, 4113)])
def test_optimizer_folded_into_the_reduction_takes_the_same_steps(d, widths, nb, B, monkeypatch):
    """hint_chain_backward_adam (the captured step of a one-process trainer) against the separate reduction + optimizer
    launches: the same weights and Adam moments bit for bit after five steps, the gradient arena left at zero"""
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(4)).to("cuda:0")
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("HINT_FUSE_ADAM", fused)
        torch.manual_seed(0)
        flow = hint_amd.HintFlow(d, nb, widths).to("cuda:0")
        tr = hint_amd.FlowTrainer(flow, use_graph=True, seed=1)           # (in-kernel noise: the same counter-based draws in both runs)
        for _ in range(5):
            tr.step(x)
        torch.cuda.synchronize()
        out[fused] = (tr.P.clone(), tr.M.clone(), tr.V.clone(), tr.G.clone())
    for a, b in zip(out["1"][:3], out["0"][:3]):
        assert torch.equal(a, b)
    assert float(out["1"][3].abs().max()) == 0.0 and float(out["0"][3].abs().max()) == 0.0
    assert not torch.equal(out["1"][0], torch.zeros_like(out["1"][0]))
