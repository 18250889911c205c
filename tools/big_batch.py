"""Large batches: index arithmetic beyond 2^31 elements (tape, workspace), the persistent tile loop.
Row independence is the check: rows of a huge batch equal the same rows processed alone.
   python tools/big_batch.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hint_amd
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000      # (one trainer at a time: 8 blocks of tape + workspace are ~120 GB here)
torch.manual_seed(0)
flow = hint_amd.HintFlow(6, 8, [140, 70, 35, 17]).to(dev)
for p in flow.parameters():
    p.data.mul_(0.3)
x = torch.randn(B, 6, device=dev)
with torch.no_grad():
    z = flow(x); J = flow.log_jacobian(run_forward=False)
    for lo in (0, B // 2 + 7, B - 1000):
        zs = flow(x[lo:lo + 1000]); Js = flow.log_jacobian(run_forward=False)
        assert torch.equal(z[lo:lo + 1000], zs) and torch.equal(J[lo:lo + 1000], Js), lo
print("forward rows independent at B =", B)
def free(t):
    del t
    import gc; gc.collect(); torch.cuda.empty_cache()

tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False, lr=0.0)
tr._check_arenas(); tr.G.zero_(); tr._fwd_bwd(x, None); torch.cuda.synchronize()
s = tr.loss_acc.sum(0)
big = [float(s[0]) / B, -float(s[1]) / B]
G = tr.G.clone()
free(tr); del tr
halves = []
for part in (x[: B // 2], x[B // 2:]):
    t2 = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False, lr=0.0)
    t2._check_arenas(); t2.G.zero_(); t2._fwd_bwd(part, None); torch.cuda.synchronize()
    s = t2.loss_acc.sum(0)
    halves.append((float(s[0]), float(s[1]), t2.G.clone(), part.shape[0]))
    free(t2); del t2
l0 = sum(h[0] for h in halves) / B; l1 = -sum(h[1] for h in halves) / B
print("losses big", big, "halves", [l0, l1])
assert abs(big[0] - l0) < 1e-4 * max(1, abs(l0)) and abs(big[1] - l1) < 1e-4 * max(1, abs(l1))
# gradients: the whole batch vs the sample-weighted sum of the halves
Gh = sum(h[2] * (h[3] / B) for h in halves)
err = ((G - Gh).abs().max() / G.abs().max()).item()
print("gradient whole vs halves rel", err)
assert err < 1e-3
print("big batch ok")
