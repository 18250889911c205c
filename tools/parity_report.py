"""Deviation of the HIP path from the CPU oracle on seeded inputs (one BASELINE-shaped block):
max |dz| / max|z|, max |dJ| / max|J|, worst relative gradient error - for comparing builds.
   [HINT_AMD_LIB=...] python tools/parity_report.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import hint_amd
from oracle import hint_oracle as orc

dev = "cuda:0"
for d, widths, dc, B in [(6, [140, 70, 35, 17], 0, 4096), (8, [128, 64, 32, 16], 0, 2048), (43, [67, 33, 16, 8], 0, 1024), (100, [224, 112, 56], 4, 256)]:
    dims_c = [(dc,)] if dc else []
    nodes = orc.build_nodes(d, dims_c, widths)
    P = orc.init_params(nodes, seed=11, scale=None)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, d, generator=gen)
    cond = [torch.randn(B, dc, generator=gen)] if dc else []
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], dims_c=dims_c, c_internal=widths)
    blk.load_state_dict({k: v.clone() for k, v in P.items()})
    blk = blk.to(dev)
    Po = {k: v.clone().double().requires_grad_(True) for k, v in P.items()}       # float64 oracle = ground truth
    xo = x.clone().double().requires_grad_(True)
    zo, Jo = orc.block_apply(nodes, Po, xo, [c.double() for c in cond], rev=False)
    (0.5 * (zo ** 2).sum(1) - Jo).mean().backward()
    xg = x.to(dev).requires_grad_(True)
    (z,) = blk([xg], c=[c.to(dev) for c in cond]); J = blk.jacobian(None)
    (0.5 * (z ** 2).sum(1) - J).mean().backward()
    ez = (z.detach().cpu().double() - zo.detach()).abs().max().item() / zo.detach().abs().max().item()
    eJ = (J.detach().cpu().double() - Jo.detach()).abs().max().item() / Jo.detach().abs().max().item()
    named = dict(blk.named_parameters())
    eg = max(((named[k].grad.cpu().double() - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)).item() for k, p in Po.items())
    ex = ((xg.grad.cpu().double() - xo.grad).abs().max() / xo.grad.abs().max()).item()
    print(f"d={d} widths={widths} dc={dc} B={B}: z {ez:.2e}  J {eJ:.2e}  dL/dx {ex:.2e}  worst dL/dW {eg:.2e}   (vs float64 oracle)")
