import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import hint_amd
from oracle import hint_oracle as orc
dev="cuda:0"
for (d, widths, B) in [(7,[3,128],2),(7,[128,128],2),(7,[3,3],2),(3,[3],2),(6,[3,16],2),(6,[16,3],2),(2,[3],2),(4,[5,16],3),(7,[16,128],2),(7,[17,128],2),(7,[3,16],2), (7,[3,32],2),(7,[3,48],2)]:
    nodes = orc.build_nodes(d, [], widths)
    P = orc.init_params(nodes, seed=1, scale=None)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(5))
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=list(widths))
    blk.load_state_dict({k: v.clone() for k, v in P.items()}); blk = blk.to(dev)
    zo, Jo = orc.block_apply(nodes, P, x, [], rev=False)
    with torch.no_grad():
        (z,) = blk([x.to(dev)]); J = blk.jacobian(None)
    print(d, widths, B, "z err", float((z.cpu()-zo).abs().max()), "J err", float((J.cpu()-Jo).abs().max()), "per-lane", (z.cpu()-zo).abs().max(0).values.numpy().round(4))
