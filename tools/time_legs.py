"""Median duration of the launches of a training step (HIP events inside un-captured steps) and of the sampling
direction, for A/B builds:  HINT_AMD_LIB=hint_amd/lib/<variant>.so python tools/time_legs.py [workload] [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hint_amd, bench

name = sys.argv[1] if len(sys.argv) > 1 else "power_hint_8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = dict(bench.WORKLOADS[name])
if os.environ.get("WIDTHS"):
    cfg["c_internal"] = [int(v) for v in os.environ["WIDTHS"].split(",")]     # (what-if runs: same tree, other hidden widths)
B = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["batch"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = hint_amd.HintFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"], max_splits=int(os.environ.get("MAX_SPLITS", "-1"))).to(dev)     # (what-if runs: a shallower tree)
with torch.no_grad():
    for p in flow.parameters():
        p.data = (0.005 * torch.randn(p.shape)).to(dev)
tr = hint_amd.FlowTrainer(flow, use_graph=False, seed=1)
x = torch.randn(B, cfg["d"], device=dev)
for _ in range(5):
    tr.step(x)
legs = [tr.timed_step(x) for _ in range(steps)]
med = {k: float(np.median([l[k] for l in legs])) for k in legs[0]}
z = torch.randn(B, cfg["d"], device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
inv, inf = [], []
from hint_amd import _lib
chain = tr._chain_infer(B) if hasattr(tr, "_chain_infer") else tr._chain_for(B)
xo = torch.empty_like(z); Jo = torch.empty(B, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(steps):
    ev[0].record()
    _lib.check(tr.lib.hint_chain_inverse(chain, z.data_ptr(), None, xo.data_ptr(), Jo.data_ptr(), None, stream), "inv")
    ev[1].record()
    torch.cuda.synchronize()
    inv.append(ev[0].elapsed_time(ev[1]) * 1e3)
med["inverse"] = float(np.median(inv))
print(os.environ.get("HINT_AMD_LIB", "default"), name, B, " ".join(f"{k.split('_kernel')[0].replace('hint_', '')}={v:.1f}" for k, v in med.items()),
      f"sum={sum(v for k, v in med.items() if k != 'inverse'):.1f} us")
