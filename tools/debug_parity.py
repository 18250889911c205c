"""Staged parity report against the CPU oracle (run on a GPU box): forward, inverse, backward of single
blocks of growing complexity; prints every deviation instead of stopping at the first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hint_amd
from oracle import hint_oracle as orc

dev = "cuda:0"
torch.manual_seed(0)
CASES = [
    (2, [16], 0, 16), (2, [20], 0, 5), (3, [40, 24], 0, 33), (6, [140, 70, 35, 17], 0, 100), (8, [128, 64, 32, 16], 0, 50),
    (6, [32, 16], 3, 40), (1, [8], 0, 7), (43, [67, 33, 16, 8], 0, 48), (100, [224, 112, 56], 4, 20), (6, [512, 256], 0, 20),
    (2, [200], 0, 20), (2, [512], 0, 20), (40, [24], 0, 20),
]
only = os.environ.get("CASE")
stages = os.environ.get("STAGES", "fib")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if b.size else 0.0


for ci, (d, widths, dc, B) in enumerate(CASES):
    if only is not None and int(only) != ci:
        continue
    dims_c = [(dc,)] if dc else []
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], dims_c=dims_c, c_internal=list(widths)).to(dev)
    x = torch.randn(B, d)
    c = [torch.randn(B, dc)] if dc else []
    nodes = orc.build_nodes(d, dims_c, widths)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    co = [t.clone().requires_grad_(True) for t in c]
    zo, Jo = orc.block_apply(nodes, P, xo, co, rev=False, clamp=4.0)
    Lo = (0.5 * (zo ** 2).sum(1) - Jo).mean()
    Lo.backward()
    msg = f"case {ci} d={d} widths={widths} dc={dc} B={B}:"
    try:
        if "f" in stages:
            with torch.no_grad():
                (z,) = blk([x.to(dev)], c=[t.to(dev) for t in c])
                J = blk.jacobian(None)
            torch.cuda.synchronize()
            msg += f" fwd z {rel(z.cpu(), zo.detach()):.2e} J {rel(J.cpu(), Jo.detach()):.2e}"
        if "i" in stages:
            with torch.no_grad():
                (xr,) = blk([zo.detach().to(dev)], c=[t.to(dev) for t in c], rev=True)
                Jr = blk.jacobian(None)
            torch.cuda.synchronize()
            msg += f" | inv x {rel(xr.cpu(), x):.2e} J {rel(Jr.cpu(), -Jo.detach()):.2e}"
        if "b" in stages:
            xg = x.to(dev).requires_grad_(True)
            cg = [t.to(dev).requires_grad_(True) for t in c]
            (z,) = blk([xg], c=cg)
            J = blk.jacobian(None)
            L = (0.5 * (z ** 2).sum(1) - J).mean()
            L.backward()
            torch.cuda.synchronize()
            msg += f" | bwd gx {rel(xg.grad.cpu(), xo.grad):.2e}"
            if dc:
                msg += f" gc {rel(cg[0].grad.cpu(), co[0].grad):.2e}"
            named = dict(blk.named_parameters())
            worst = sorted(((rel(named[k].grad.cpu(), p.grad), k) for k, p in P.items()), reverse=True)[:3]
            msg += " gW " + " ".join(f"{k}:{e:.1e}" for e, k in worst)
    except Exception as e:  # noqa: BLE001
        msg += f" EXCEPTION {type(e).__name__}: {e}"
    print(msg, flush=True)
