"""time the block kernels of one config through the C ABI (uses HINT_AMD_LIB if set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hint_amd, bench
d, widths, B = 6, [140, 70, 35, 17], 4096
if len(sys.argv) > 1:
    d = int(sys.argv[1]); widths = [int(v) for v in sys.argv[2].split(",")]; B = int(sys.argv[3])
dev = torch.device("cuda:0")
flow = hint_amd.HintFlow(d, 1, widths).to(dev)
tr = hint_amd.FlowTrainer(flow, use_graph=False)
x = torch.randn(B, d, device=dev)
legs = bench.kernel_legs(tr, x, reps=100)
print(os.environ.get("HINT_AMD_LIB", "default"), {k: round(v, 2) for k, v in legs.items()})
