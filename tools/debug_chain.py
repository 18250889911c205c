import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hint_amd
torch.manual_seed(0)
dev = "cuda:0"
d, B = 6, 64
flow = hint_amd.HintFlow(d, 2, [32, 16]).to(dev)
x = torch.randn(B, d, device=dev)
with torch.no_grad():
    z_ref = flow(x); J_ref = flow.log_jacobian(run_forward=False)
    # block 0 then block 1 via chain API
    e0, e1 = flow.blocks[0].tree._engine, flow.blocks[1].tree._engine
    la = torch.zeros(2, device=dev)
    h0, J0, t0 = e0.forward_chain(x, None, None, None, None, True)
    (h0_ref,) = flow.blocks[0]([x]); J0_ref = flow.blocks[0].jacobian(None)
    print("blk0 z err", (h0 - h0_ref).abs().max().item(), "J err", (J0 - J0_ref).abs().max().item())
    W = flow.perms[1].W
    h1, J1, t1 = e1.forward_chain(h0, None, W, J0, la, True)
    print("chain z err", (h1 - z_ref).abs().max().item(), "J err", (J1 - J_ref).abs().max().item())
    xp = t1[(e1.lib.hint_plan_tape_floats(e1.plan, B) // (B * d) - 1) * B * d:].view(B, d)
    print("perm err", (xp - h0 @ W).abs().max().item())
    print("loss_acc", la.tolist(), (0.5 * (z_ref ** 2).sum()).item(), J_ref.sum().item())
