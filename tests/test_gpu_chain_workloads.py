"""Chain-level parity at the BASELINE workloads' shapes (cfg 3 = gas_hint_8, cfg 4 =
the d = 100 lane of conditional_hint_4_full.py with and without a condition, cfg 5 = miniboone_hint_10, and
the *_big width h = 512): the whole flow per launch (hint_chain_forward / hint_chain_backward /
hint_chain_inverse) against oracle/hint_oracle.py's OracleFlow evaluated in float64 on the same weights and
rows.  Bars: NLL within 1e-4 relative (north_star), the flat gradient within 1e-4 relative (norm-wise),
inverse round trip within 1e-4 absolute.  Rows with a hidden pre-activation within KINK = 5e-7 (of its layer's
largest one in that row) of zero in the float64 oracle are left out beforehand (and counted): float32 summation order decides on which side of the ReLU
kink such a row lands, and either subgradient is a correct answer (tools/fuzz_parity.py shows the effect on
single rows).  Plus the determinism of the weight gradients (runs bit-identical)."""
import math

import numpy as np
import pytest
import torch

import hint_amd
from oracle import hint_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

WORKLOADS = [
    # name, d, dc, n_blocks, widths, rows, weight scale (randn * scale: log-dets of a few nats, z of order 1..10 -
    # the reference's own 0.005 (train_unconditional.py:165-167) would leave s and t next to zero)
    ("cfg2_power_hint_8", 6, 0, 8, [140, 70, 35, 17], 1024, 0.06),
    ("cfg3_gas_hint_8", 8, 0, 8, [128, 64, 32, 16], 512, 0.06),
    ("cfg4_plus_x_lane", 100, 0, 4, [224, 112, 56], 160, 0.03),
    ("cfg4_plus_x_lane_cond", 100, 4, 4, [224, 112, 56], 96, 0.03),
    ("cfg5_miniboone_hint_10", 43, 0, 10, [67, 33, 16, 8], 200, 0.06),
    ("plus_hint_4_big", 100, 0, 2, [512, 256, 128, 64], 48, 0.03),
    # full BASELINE batch sizes above 16 rows x CUs: the plan variant for more row tiles than CUs
    # (configs/uci_data/gas_hint_8.py:29-36,55-71 at its batch of 8192; power_hint_8 at the same size)
    ("cfg3_gas_hint_8_B8192", 8, 0, 8, [128, 64, 32, 16], 8192, 0.06),
    ("cfg2_power_hint_8_B8192", 6, 0, 8, [140, 70, 35, 17], 8192, 0.06),
    # gradients at the batch sizes the general plans really run at: more row tiles than the chip has CUs (273 tiles, the last
    # one ragged: the persistent tile loop of hint_apply_kernel / hint_bwd_kernel_fly and part B's row splits at d = 100 - the
    # reference's own batch for cfg 4 is 10 000 rows, configs/plus_shape/conditional_hint_4_full.py:37), with and without a
    # condition, and cfg 5's whole chain at its per-GPU batch (configs/uci_data/miniboone_hint_8.py; BASELINE cfg 5: 4096 rows per GPU)
    ("cfg4_plus_x_lane_B4368", 100, 0, 2, [224, 112, 56], 4368, 0.03),
    ("cfg4_plus_x_lane_cond_B4368", 100, 4, 2, [224, 112, 56], 4368, 0.03),
    ("cfg5_miniboone_hint_10_B4096", 43, 0, 10, [67, 33, 16, 8], 4096, 0.06),
    # BASELINE cfg 1 at its exact shape: configs/uci_data/power_hint_4.py:29-30,66 - 4 blocks [200,100,50,25], batch 512
    ("cfg1_power_hint_4_B512", 6, 0, 4, [200, 100, 50, 25], 512, 0.06),
]
LARGE = {"cfg4_plus_x_lane_B4368", "cfg4_plus_x_lane_cond_B4368", "cfg5_miniboone_hint_10_B4096"}
# at most this share of a batch may be left out as "next to a ReLU kink": per workload the share observed on MI355X
# (deterministic: seeded inputs, the float64 oracle decides) + 3 points; the counts of a run go to gpurun_out/kink_rows.json
MAX_KINK_ROWS = {
    # observed (round 4, gpurun_out/kink_rows.json): 9/1024, 7/512, 16/160, 3/96, 3/200, 2/48, 96/8192, 78/8192
    "cfg2_power_hint_8": 0.04, "cfg3_gas_hint_8": 0.045, "cfg4_plus_x_lane": 0.13, "cfg4_plus_x_lane_cond": 0.065,
    "cfg5_miniboone_hint_10": 0.045, "plus_hint_4_big": 0.075, "cfg3_gas_hint_8_B8192": 0.042, "cfg2_power_hint_8_B8192": 0.04,
    # (round 5, decided by the float64 oracle on the CPU: 138/4368, 156/4368, 64/4096)
    "cfg4_plus_x_lane_B4368": 0.062, "cfg4_plus_x_lane_cond_B4368": 0.066, "cfg5_miniboone_hint_10_B4096": 0.046,
    "cfg1_power_hint_4_B512": 0.045,
}


def record_kink_rows(name, dropped, B):
    """keep the dropped-row counts where a reader of the run finds them (pytest -q swallows prints)"""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        f = os.path.join(out, "kink_rows.json")
        have = json.load(open(f)) if os.path.exists(f) else {}
        have[name] = {"dropped": dropped, "rows": B, "share": dropped / B, "cap": MAX_KINK_ROWS[name], "kink": KINK}
        json.dump(have, open(f, "w"), indent=1)
    except OSError:
        pass


def make_pair(d, dc, n_blocks, widths, scale, seed=0):
    """OracleFlow in float64 and the same flow (same float32 weights) on the GPU"""
    dims_c = [(dc,)] if dc else ()
    ref = orc.OracleFlow(d, n_blocks, widths, dims_c=dims_c, seed=seed, init_scale=scale, dtype=torch.float64)
    flow = hint_amd.HintFlow(d, n_blocks, widths, ndim_c=dc)
    for i, blk in enumerate(flow.blocks):
        blk.load_state_dict({k: v.float() for k, v in ref.params[i].items()})
        if ref.perms[i] is not None:
            flow.perms[i].W.copy_(ref.perms[i].float())
        # the oracle sees exactly the float32 weights the GPU has
        ref.params[i] = {k: v.float().double() for k, v in ref.params[i].items()}
    ref.perms = [None if p is None else p.float().double() for p in ref.perms]
    return ref, flow.to(DEV)


KINK = 5e-7


def rows_off_the_kinks(ref, x64, cr, out=None):
    """mask of the rows none of whose hidden pre-activations (whole chain, float64) lies within KINK of zero; out (a list)
    receives the oracle's (z, J) of ALL rows - a row next to a kink has an ambiguous subgradient, not an ambiguous forward"""
    dist = torch.full((x64.shape[0],), float("inf"), dtype=torch.float64)
    relu = torch.relu

    def spy(t):
        nonlocal dist
        if t.numel() > 0:
            a = t.detach().abs().reshape(t.shape[0], -1)
            dist = torch.minimum(dist, a.min(dim=1).values / a.max(dim=1).values.clamp(min=1e-3))
        return relu(t)
    torch.relu = spy
    try:
        with torch.no_grad():
            zJ = ref.forward(x64, cr)
            if out is not None:
                out.append(zJ)
    finally:
        torch.relu = relu
    return dist > KINK


def flat_grads(tr, flow):
    out = {}
    for bi, ((a, b), eng) in enumerate(zip(tr.slices, tr.engines)):
        for p, g in zip(eng.params, eng.split_flat(tr.G[a:b])):
            name = [n for n, q in flow.blocks[bi].named_parameters() if q is p][0]
            out[(bi, name)] = g.detach().double().cpu()
    return out


@pytest.mark.parametrize("name,d,dc,n_blocks,widths,B,scale",
                         [pytest.param(*w, marks=pytest.mark.timeout(900)) if w[0] in LARGE else w for w in WORKLOADS],
                         ids=[w[0] for w in WORKLOADS])
def test_chain_nll_gradient_and_inverse_match_oracle(name, d, dc, n_blocks, widths, B, scale):
    torch.set_num_threads(min(16, torch.get_num_threads()))      # (the float64 oracle: a GPU box has 256 host cores)
    ref, flow = make_pair(d, dc, n_blocks, widths, scale)
    g = torch.Generator().manual_seed(7)
    x64 = torch.randn(B, d, generator=g, dtype=torch.float64).float().double()
    c64 = torch.randn(B, dc, generator=g, dtype=torch.float64).float().double() if dc else None
    cr = (c64,) if dc else ()
    all_rows = []
    keep = rows_off_the_kinks(ref, x64, cr, out=all_rows)
    dropped = B - int(keep.sum())
    print(f"{name}: {dropped} of {B} rows left out (pre-activation within {KINK} of a ReLU kink)")
    record_kink_rows(name, dropped, B)
    assert dropped <= MAX_KINK_ROWS[name] * B, f"{dropped} of {B} rows next to a ReLU kink (cap {MAX_KINK_ROWS[name]:.0%})"
    # the exclusion waives the SUBGRADIENT choice only: z and log-det of every row - the dropped ones too - are checked here,
    # on the chained training forward (tape and all) at the full batch
    z_all, J_all = all_rows[0]
    tr0 = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    tr0._check_arenas()
    tr0._pack_all()
    chain = tr0._chain_for(B)
    xa = x64.float().to(DEV)
    ca = c64.float().to(DEV) if dc else None
    zg, Jg = torch.empty_like(xa), torch.empty(B, device=DEV)
    from hint_amd import _lib
    _lib.check(tr0.lib.hint_chain_forward(chain, xa.data_ptr(), ca.data_ptr() if dc else None, zg.data_ptr(), Jg.data_ptr(), None,
                                          None, torch.cuda.current_stream().cuda_stream), "hint_chain_forward")
    zs = max(1.0, float(z_all.abs().max()))
    dz = (zg.double().cpu() - z_all).abs().max(dim=1).values
    dJ = (Jg.double().cpu() - J_all).abs()
    assert float(dz.max()) <= 1e-4 * zs and float(dJ.max()) <= 1e-4 * max(1.0, float(J_all.abs().max())), \
        (name, float(dz.max()), float(dJ.max()))
    if dropped:
        print(f"{name}: dropped rows' forward: |dz| <= {float(dz[~keep].max()):.2e}, |dJ| <= {float(dJ[~keep].max()):.2e}")
    del tr0
    x64 = x64[keep]
    if dc:
        c64 = c64[keep]
        cr = (c64,)
    B = x64.shape[0]
    for p in ref.parameters():
        p.requires_grad_(True)
    z_ref, J_ref = ref.forward(x64, cr)
    l0, l1 = ref.loss_terms(z_ref, J_ref)
    (l0 + l1).backward()
    nll_ref = float(l0 + l1) + 0.5 * d * math.log(2 * math.pi)

    x = x64.float().to(DEV)
    c = c64.float().to(DEV) if dc else None
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    assert tr._chainable
    tr._check_arenas()
    tr.G.zero_()
    tr._fwd_bwd(x, c)
    torch.cuda.synchronize()
    s = tr.loss_acc.double().sum(dim=0).cpu()
    nll = float(s[0] / B - s[1] / B) + 0.5 * d * math.log(2 * math.pi)
    assert abs(nll - nll_ref) <= 1e-4 * abs(nll_ref), (nll, nll_ref)

    grads = flat_grads(tr, flow)
    num = den = 0.0
    gmax = max(float(ref.params[bi][k].grad.abs().max()) for (bi, k) in grads)
    for (bi, k), gg in grads.items():
        r = ref.params[bi][k].grad
        num += float(((gg - r) ** 2).sum()); den += float((r ** 2).sum())
        # per tensor (a wrong small tensor must not hide in the pooled norm): 1e-4 of the tensor's own norm, with a
        # floor of 1e-6 of the model's largest gradient entry per element for tensors whose gradient is all but zero
        tn, tr_ = float(((gg - r) ** 2).sum()) ** 0.5, float((r ** 2).sum()) ** 0.5
        assert tn <= 1e-4 * tr_ + 1e-6 * gmax * math.sqrt(r.numel()), (name, bi, k, tn, tr_)
    assert math.sqrt(num / den) <= 1e-4, (name, math.sqrt(num / den))

    # sampling direction: the whole chain in one launch
    with torch.no_grad():
        z = flow(x, c=c)
        xr, Jr = tr.sample(z, c)
        zi = torch.randn(B, d, generator=g, dtype=torch.float64).float()
        xs, Js = tr.sample(zi.to(DEV), c)
        xs_ref, Js_ref = ref.inverse(zi.double(), cr)
    assert (xr - x).abs().max().item() < 1e-4 * max(1.0, x.abs().max().item())
    np.testing.assert_allclose(Jr.double().cpu().numpy(), -J_ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    scale = max(1.0, float(xs_ref.abs().max()))
    assert float((xs.double().cpu() - xs_ref).abs().max()) < 1e-4 * scale
    np.testing.assert_allclose(Js.double().cpu().numpy(), Js_ref.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("d,dc,n_blocks,widths,B", [
    (6, 0, 8, [140, 70, 35, 17], 4096),            # the bench workload: 64 row splits of 4096 rows
    (8, 3, 3, [64, 32, 16], 1000),
    (43, 0, 2, [67, 33, 16, 8], 333),
])
def test_weight_gradients_are_deterministic(d, dc, n_blocks, widths, B):
    """the K-split of the weight-gradient GEMMs is summed in a fixed order (slabs + hint_wreduce_kernel, no
    atomics): two backward passes over the same tape give bit-identical gradients, overwrite and accumulate"""
    torch.manual_seed(3)
    flow = hint_amd.HintFlow(d, n_blocks, widths, ndim_c=dc).to(DEV)
    x = torch.randn(B, d, device=DEV)
    c = torch.randn(B, dc, device=DEV) if dc else None
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    tr._check_arenas()
    runs = []
    for _ in range(3):
        tr.G.zero_()
        tr._fwd_bwd(x, c)
        torch.cuda.synchronize()
        runs.append(tr.G.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2])
    assert runs[0].abs().sum().item() > 0


@pytest.mark.parametrize("d,dc,n_blocks,widths,B", [(43, 0, 3, [67, 33, 16, 8], 200), (100, 0, 2, [224, 112, 56], 96),
                                                    (100, 4, 2, [224, 112, 56], 48)])
def test_l2_prefetch_changes_no_result(d, dc, n_blocks, widths, B):
    """the general kernels' L2 warm-up (hint_device.hpp prefetch_consumer: loads into an LDS sink nobody reads, two
    consumers ahead) is a hint: losses, z, log-dets and gradients with it (default) and without it
    (hint_debug_set_prefetch(0)) are bit-identical - subtree plan (cfg 5 shape), general plan (cfg 4's x lane), with a
    condition.  That the switch reaches BOTH row kernels is read off the launches' LDS sizes: the sink is 256 bytes
    behind everything else (round 4's backward launch never had one: its warm-up was dead code)."""
    from hint_amd import _lib
    lib = _lib.load()
    torch.manual_seed(11)
    flow = hint_amd.HintFlow(d, n_blocks, widths, ndim_c=dc).to(DEV)
    with torch.no_grad():
        for p in flow.parameters():
            p.copy_(0.04 * torch.randn_like(p))
    x = torch.randn(B, d, device=DEV)
    c = torch.randn(B, dc, device=DEV) if dc else None
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    tr._check_arenas()
    outs, lds = [], []
    before = lib.hint_debug_set_prefetch(1)
    try:
        for pf in (0, 1, 0):
            lib.hint_debug_set_prefetch(pf)
            tr.G.zero_()
            tr._fwd_bwd(x, c)
            torch.cuda.synchronize()
            outs.append([tr.G.clone(), tr.loss_acc.clone()])      # flat gradient; the two loss sums (0.5 |z|^2, log-det) of the forward
            lds.append((lib.hint_debug_last_lds_bytes(0), lib.hint_debug_last_lds_bytes(1)))
    finally:
        lib.hint_debug_set_prefetch(before)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][0]).all() and outs[0][0].abs().sum().item() > 0
    assert lds[0] == lds[2]
    # (a plan that fills the 160 KiB has no room for the sink: then the sizes agree and nothing was switched)
    for k in (0, 1):
        assert lds[1][k] in (lds[0][k], lds[0][k] + 256), lds
    assert lds[1][0] == lds[0][0] + 256 and lds[1][1] == lds[0][1] + 256, lds
