"""Subtree groups (hint_sub.hpp) against the float32 oracle on single blocks: forward, inverse and gradients.
   python tools/sub_debug.py [fwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hint_amd
from oracle import hint_oracle as orc
dev = "cuda:0"
only_fwd = len(sys.argv) > 1 and sys.argv[1] == "fwd"
for (d, widths, B) in [(43, [67, 33, 16, 8], 37), (43, [67, 33, 16, 8], 4100), (20, [40, 16, 8], 16), (33, [20, 30, 12], 5), (64, [16, 16, 16, 16], 70), (100, [48, 24, 20, 12, 8, 8], 33)]:
    nodes = orc.build_nodes(d, [], widths)
    P = orc.init_params(nodes, seed=1, scale=None)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(5))
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=list(widths))
    blk.load_state_dict({k: v.clone() for k, v in P.items()}); blk = blk.to(dev)
    zo, Jo = orc.block_apply(nodes, P, x, [], rev=False)
    with torch.no_grad():
        (z,) = blk([x.to(dev)]); J = blk.jacobian(None)
        (xr,) = blk([z], rev=True); Jr = blk.jacobian(None, rev=True)
    print(d, widths, B, "z err %.2e" % float((z.cpu() - zo).abs().max()), "J err %.2e" % float((J.cpu() - Jo).abs().max()),
          "round trip %.2e" % float((xr.cpu() - x).abs().max()), "J+Jr %.2e" % float((J + Jr).abs().max()))
    if only_fwd:
        continue
    # gradients of a scalar loss against the oracle's autograd
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xg = x.clone().requires_grad_(True)
    zo, Jo = orc.block_apply(nodes, Pg, xg, [], rev=False)
    gz = torch.randn(B, d, generator=torch.Generator().manual_seed(6)); gJ = torch.randn(B, generator=torch.Generator().manual_seed(7))
    ((zo * gz).sum() + (Jo * gJ).sum()).backward()
    xd = x.to(dev).requires_grad_(True)
    for p in blk.parameters():
        p.grad = None
    (z,) = blk([xd]); J = blk.jacobian(None)
    ((z * gz.to(dev)).sum() + (J * gJ.to(dev)).sum()).backward()
    worst = 0.0; wname = ""
    sd = dict(blk.named_parameters())
    for k, v in Pg.items():
        g = sd[k].grad.cpu(); ref = v.grad
        e = float((g - ref).abs().max()) / max(1e-6, float(ref.abs().max()))
        if e > worst: worst, wname = e, k
    print("     gx err %.2e" % (float((xd.grad.cpu() - xg.grad).abs().max()) / float(xg.grad.abs().max())), "worst parameter gradient (relative to its max) %.2e %s" % (worst, wname))
