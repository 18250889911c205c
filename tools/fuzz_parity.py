"""Randomised parity sweep (GPU): random block shapes - lanes, hidden widths, conditions, batch sizes,
max_splits / min_split_size - through forward, inverse and backward against the CPU oracle in float32
(the arithmetic the reference runs in: against float64 a single row whose ReLU pre-activation sits at
zero flips its subgradient and moves a weight gradient by 1e-2, in the oracle exactly as on the GPU).

Rows on a ReLU kink: when forward and log-det agree to 1e-5 but a gradient does not, the rows whose input
gradient deviates are looked up in the oracle: a row is ACCEPTED as a kink row only if one of its hidden
pre-activations lies within KINK_EPS = 1e-5 (x the row's largest pre-activation, at least 1) of zero - the
distance float32 summation order can move it across.  Such rows are printed (index, the pre-activation,
its deviation) and dropped, at most KINK_ROWS = 3 per case; anything else fails the sweep.
   python tools/fuzz_parity.py [n_cases] [seed] [wide | deep | lean]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hint_amd
from oracle import hint_oracle as orc

KINK_EPS, KINK_ROWS = 1e-5, 3


def kink_distance(nodes, P, x, cond, clamp):
    """per row: the hidden pre-activation of the oracle closest to zero, relative to the row's largest one"""
    pre = []
    relu = torch.relu

    def spy(t):
        pre.append(t.detach().abs())
        return relu(t)
    torch.relu = spy
    try:
        with torch.no_grad():
            orc.block_apply(nodes, P, x, cond, rev=False, clamp=clamp)
    finally:
        torch.relu = relu
    allp = torch.cat([p.reshape(p.shape[0], -1) for p in pre if p.numel() > 0], dim=1)
    return allp.min(dim=1).values / allp.max(dim=1).values.clamp(min=1.0)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
worst = dict(z=0.0, J=0.0, gx=0.0, gw=0.0, rt=0.0)
for case in range(n_cases):
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"       # up to the limits: 128 lanes, split (h > 384) nodes
    d = rng.choice([2, 6, 33, 100, 127, 128] if wide else [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 21, 31, 43, 64])
    depth = rng.randint(1, 3 if wide else 4)
    widths = [rng.choice([200, 224, 256, 385, 400, 512] if wide else [3, 8, 15, 16, 17, 24, 33, 48, 64, 70, 100, 128, 140])
              for _ in range(depth)]
    dc = rng.choice([0, 0, 0, 1, 3, 5])
    B = rng.choice([1, 17, 100, 257] if wide else [1, 2, 15, 16, 17, 33, 100, 257, 1000])
    if len(sys.argv) > 3 and sys.argv[3] == "deep":
        # deep trees with narrow nets at the bottom: the subtree groups of the general kernels (hint_sub.hpp), all tree shapes
        d = rng.choice([20, 26, 33, 43, 50, 64, 77, 100, 128])
        depth = rng.randint(3, 7)
        widths = [rng.choice([16, 24, 33, 48, 67]) for _ in range(rng.randint(1, 2))] + [rng.choice([3, 8, 12, 16]) for _ in range(depth - 1)]
        dc = rng.choice([0, 0, 0, 0, 2])
        B = rng.choice([1, 16, 33, 257, 4112])
    if len(sys.argv) > 3 and sys.argv[3] == "lean":
        # wide trees whose deep levels are general groups of small nodes with 17..100 hidden units: the kernel instances whose rows
        # make the thin layers themselves (hint_rows.hpp row_body FLY: hint_apply_kernel<REV, true>, hint_bwd_kernel_fly), odd
        # tile counts, units of more than one row, the backward's active-lane boundaries
        d = rng.choice([20, 26, 33, 43, 50, 64, 77, 100, 128])
        depth = rng.randint(1, 3)
        widths = [rng.choice([17, 24, 33, 48, 56, 64, 70, 100]) for _ in range(depth)]
        dc = rng.choice([0, 0, 0, 0, 3])
        B = rng.choice([1, 16, 33, 257, 1000])
    if not wide and d <= 16 and case % 10 == 7:
        B = rng.choice([4112, 6000])      # more row tiles than CUs: the launch picks the plan variant for large batches
    max_splits = rng.choice([-1, -1, 0, 1, 2])
    min_split = rng.choice([2, 2, 3])
    clamp = rng.choice([4.0, 4.0, 2.0])
    dims_c = [(dc,)] if dc else []
    try:
        nodes = orc.build_nodes(d, dims_c, widths, max_splits=max_splits, min_split_size=min_split)
    except TypeError:
        nodes = orc.build_nodes(d, dims_c, widths)
        max_splits, min_split = -1, 2
    P = orc.init_params(nodes, seed=case, scale=None)
    gen = torch.Generator().manual_seed(1000 + case)
    x = torch.randn(B, d, generator=gen)
    cond = [torch.randn(B, dc, generator=gen)] if dc else []
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], dims_c=dims_c, c_internal=widths, clamp=clamp,
                                                   max_splits=max_splits, min_split_size=min_split)
    blk.load_state_dict({k: v.clone() for k, v in P.items()})
    blk = blk.to(dev)
    sc = lambda t: max(1.0, t.detach().abs().max().item())

    def compare(x, cond):
        Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        xo = x.clone().requires_grad_(True)
        zo, Jo = orc.block_apply(nodes, Po, xo, cond, rev=False, clamp=clamp)
        (0.5 * (zo ** 2).sum(1) - Jo).mean().backward()
        blk.zero_grad()
        xg = x.to(dev).requires_grad_(True)
        cg = [c.to(dev) for c in cond]
        (z,) = blk([xg], c=cg); J = blk.jacobian(None)
        (0.5 * (z ** 2).sum(1) - J).mean().backward()
        with torch.no_grad():
            (xr,) = blk([z.detach()], c=cg, rev=True)
        named = dict(blk.named_parameters())
        row_err = (xg.grad.cpu() - xo.grad).abs().max(dim=1).values / (xo.grad.abs().max() + 1e-30)
        return dict(z=(z.detach().cpu() - zo.detach()).abs().max().item() / sc(zo),
                    J=(J.detach().cpu() - Jo.detach()).abs().max().item() / sc(Jo),
                    gx=row_err.max().item(),
                    # weight gradients: largest deviation in any tensor, relative to the largest gradient entry of
                    # the block (per tensor, a tiny bias gradient makes one kink row look like a 1e-2 error)
                    gw=max([(named[k].grad.cpu() - p.grad).abs().max().item()
                            for k, p in Po.items() if p.grad is not None and p.numel() > 0] + [0.0])
                       / max([p.grad.abs().max().item() for p in Po.values() if p.grad is not None and p.numel() > 0] + [1e-30]),
                    rt=(xr.cpu() - x).abs().max().item() / sc(x)), row_err

    try:
        e, row_err = compare(x, cond)
    except hint_amd.HintAmdError as err:           # the documented limit: one net of a node must fit the 160 KiB LDS
        if "LDS" in str(err):
            print(f"limit case {case}: d={d} widths={widths} dc={dc}: {str(err)[-80:]}", flush=True)
            continue
        raise
    note = ""
    if (e["gx"] > 1e-4 or e["gw"] > 2e-4) and e["z"] <= 1e-5 and e["J"] <= 1e-5:
        # a row whose ReLU pre-activation rounds to the other side of zero on the GPU than in the oracle gets
        # another subgradient: drop the (few) rows that differ and look again
        keep = row_err <= 3e-5
        if 0 < int((~keep).sum()) <= KINK_ROWS and int(keep.sum()) > 0:
            dist = kink_distance(nodes, P, x, cond, clamp)
            rows = [int(i) for i in torch.nonzero(~keep).flatten()]
            print(f"    case {case}: rows off in g_x: " + ", ".join(
                f"row {i} (g_x dev {row_err[i].item():.1e}, nearest pre-activation {dist[i].item():.1e})" for i in rows), flush=True)
            if all(dist[i].item() <= KINK_EPS for i in rows):
                note = f" [{len(rows)} row(s) on a ReLU kink dropped: {rows}]"
                e, row_err = compare(x[keep], [c[keep] for c in cond])
            else:
                note = " [deviating rows are NOT on a ReLU kink]"
    if e["rt"] > 1e-3:
        # an ill-conditioned block (exp(a) spans several orders of magnitude): what does the float32 ORACLE's own round trip do, and
        # does the GPU's inverse of the oracle's z agree with the oracle's inverse of it?
        with torch.no_grad():
            zo, _ = orc.block_apply(nodes, {k: v.clone() for k, v in P.items()}, x, cond, rev=False, clamp=clamp)
            xro, _ = orc.block_apply(nodes, {k: v.clone() for k, v in P.items()}, zo, cond, rev=True, clamp=clamp)
            (xrg,) = blk([zo.to(dev)], c=[c.to(dev) for c in cond], rev=True)
        rt_o = (xro - x).abs().max().item() / sc(x)
        inv_dev = (xrg.cpu() - xro).abs().max().item() / sc(x)
        print(f"    case {case}: round trip {e['rt']:.1e} - the oracle's own {rt_o:.1e}; GPU inverse vs oracle inverse of the same z {inv_dev:.1e}", flush=True)
        note += f" [ill-conditioned: oracle round trip {rt_o:.1e}]"
    bad = e["z"] > 1e-5 or e["J"] > 1e-5 or e["gx"] > 1e-4 or e["gw"] > 2e-4
    for k in worst: worst[k] = max(worst[k], e[k])
    if bad or note or case % 10 == 0:
        print(("BAD " if bad else "ok  ") + note + f"case {case}: d={d} widths={widths} dc={dc} B={B} max_splits={max_splits} min_split={min_split} clamp={clamp} "
              + " ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
    assert not bad
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()})
