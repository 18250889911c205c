import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import hint_amd
from oracle import hint_oracle as orc
torch.manual_seed(0)
dev='cuda:0'
for h in (16, 17, 32, 48, 64, 80):
    d=2
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=[h]).to(dev)
    x = torch.randn(16, d)
    nodes = orc.build_nodes(d, (), [h])
    for zero_hi in (False, True):
        with torch.no_grad():
            if zero_hi:
                for net in (blk.tree.s, blk.tree.t):
                    net[2].weight[:, 16:] = 0
                    net[4].weight[:, 16:] = 0
            P = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items()}
            zo, Jo = orc.block_apply(nodes, P, x, (), rev=False)
            (z,) = blk([x.to(dev)])
        err = (z.cpu()-zo).abs().max().item()
        print(f"h={h} zero_hi={zero_hi} max err {err:.3e}")
