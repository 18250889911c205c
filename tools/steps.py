"""Run N training steps of a bench workload and nothing else (driver for rocprofv3 passes: every
launch of the block kernels in this process is one of the step's launches).
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/steps.py [workload] [steps]
workload: a key of bench.WORKLOADS (chained launches of FlowTrainer) or of bench.CONDITIONAL (ConditionalFlowTrainer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hint_amd, bench

name = sys.argv[1] if len(sys.argv) > 1 else "power_hint_8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
if name in bench.CONDITIONAL:
    cfg = bench.CONDITIONAL[name]
    model = hint_amd.ConditionalHintFlow(cfg["nx"], cfg["ny"], cfg["n_blocks"], cfg["hidden"]).to(dev)
    with torch.no_grad():
        for p in model.parameters():
            p.data = (0.005 * torch.randn(p.shape)).to(dev)
    tr = hint_amd.ConditionalFlowTrainer(model, use_graph=False)     # no graph: the profiler sees plain launches
    x = torch.randn(cfg["batch"], cfg["nx"], device=dev)
    y = torch.randn(cfg["batch"], cfg["ny"], device=dev)
    for _ in range(steps):
        out = tr.step(x, y)
    torch.cuda.synchronize()
    print("ok", name, steps, [float(v) for v in out])
    sys.exit(0)
cfg = bench.WORKLOADS[name]
flow = hint_amd.HintFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"]).to(dev)
with torch.no_grad():
    for p in flow.parameters():
        p.data = (0.005 * torch.randn(p.shape)).to(dev)
tr = hint_amd.FlowTrainer(flow, use_graph=False, seed=1)     # no graph: the profiler sees plain launches
x = torch.randn(cfg["batch"], cfg["d"], device=dev)
for _ in range(steps):
    tr.step(x)
z = torch.randn(cfg["batch"], cfg["d"], device=dev)
for _ in range(10):                                           # and the sampling direction (hint_apply_kernel<true>)
    tr.sample(z)
torch.cuda.synchronize()
print("ok", name, steps, [float(v) for v in tr.last_losses()])
