import os, sys
sys.path.insert(0, os.getcwd())
import torch, hint_amd
dev = "cuda:0"
torch.manual_seed(0)
for d, widths, nb, B in [(1, [8], 2, 33), (2, [8, 4], 3, 1), (128, [32, 16], 2, 50), (6, [140, 70, 35, 17], 8, 4096), (5, [9], 1, 16)]:
    flow = hint_amd.HintFlow(d, nb, widths).to(dev)
    for p in flow.parameters():
        # (torch's default init makes deep trees / long chains expand by many orders of magnitude)
        p.data.mul_(0.1 if (d >= 100 or nb >= 8) else 0.5)
    x = torch.randn(B, d, device=dev)
    # autograd path
    z = flow(x); J = flow.log_jacobian(run_forward=False)
    (0.5 * (z ** 2).sum(1).mean() - J.mean()).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in flow.parameters()]).clone()
    with torch.no_grad():
        xr = flow(z.detach(), rev=True)
    rt = (xr - x).abs().max().item()
    for use_graph in (False, True):
        tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=use_graph, lr=0.0)   # lr 0: weights stay put
        l0, l1 = tr.step(x)
        l0b, l1b = tr.step(x)
        assert abs(float(l0) - float(l0b)) < 1e-5 * max(1, abs(float(l0))), (d, float(l0), float(l0b))
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    tr._check_arenas(); tr.G.zero_(); tr._fwd_bwd(x, None); torch.cuda.synchronize()
    fast = torch.cat([g.reshape(-1) for e, (a, b) in zip(tr.engines, tr.slices) for g in e.split_flat(tr.G[a:b])])
    err = (fast - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    print(f"d={d} widths={widths} blocks={nb} B={B}: round trip {rt:.2e}, grad fast-vs-autograd rel {err:.2e}, loss {float(l0) + float(l1):.4f}")
    assert rt < 1e-3 and err < 1e-4
print("edge cases ok")
