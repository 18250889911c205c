import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hint_amd, bench
dev = torch.device("cuda:0")
flow = hint_amd.HintFlow(6, 8, [140, 70, 35, 17]).to(dev)
tr = hint_amd.FlowTrainer(flow, use_graph=False)
x = torch.randn(4096, 6, device=dev)
legs = bench.kernel_legs(tr, x, reps=100)
print(os.environ.get("HINT_DW_SPLITS"), {k: round(v, 1) for k, v in legs.items() if "dw" in k})
