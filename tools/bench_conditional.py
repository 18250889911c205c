"""Training-step rate of the conditional two-lane model (conditional_hint_4_full.py: x d=100, y d=4,
4 blocks, h=224) through the module path: autograd over the HIP node kernels, torch.optim.Adam,
per-parameter clamp - the statements of train_conditional.py:120-150 as they stand.
   python tools/bench_conditional.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hint_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = hint_amd.ConditionalHintFlow(100, 4, 4, 224).to(dev)
with torch.no_grad():
    for p in model.parameters():
        p.data = 0.005 * torch.randn_like(p)
params = [p for p in model.parameters() if p.requires_grad]
optim = torch.optim.Adam(params, lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
x = torch.randn(B, 100, device=dev); y = torch.randn(B, 4, device=dev)


def step():
    optim.zero_grad()
    xn = x + 0.01 * torch.randn_like(x)
    z_y, z_x = model([y, xn])
    z = torch.cat([z_x, z_y], dim=-1)
    log_jacobian = model.log_jacobian(run_forward=False)
    loss = 0.5 * torch.sum(z ** 2, dim=1).mean() - log_jacobian.mean()
    loss.backward()
    for p in params:
        p.grad.data.clamp_(-5.00, 5.00)
    optim.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"conditional_hint_4_full (x 100, y 4, 4 blocks, h 224), batch {B}: {B / dt:,.0f} samples/s, {dt * 1e3:.2f} ms/step, "
      f"loss {loss.item():.4f}, params {sum(p.numel() for p in params):,}")

# where the time goes (synchronised sections)
def timed(f, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r

def fwd():
    z_y, z_x = model([y, x])
    return 0.5 * torch.sum(torch.cat([z_x, z_y], dim=-1) ** 2, dim=1).mean() - model.log_jacobian(run_forward=False).mean()

t_f, loss = timed(fwd)
optim.zero_grad()
t_b, _ = timed(lambda: fwd().backward())
t_c, _ = timed(lambda: [p.grad.data.clamp_(-5.0, 5.0) for p in params])
t_o, _ = timed(optim.step)
with torch.no_grad():
    t_n, _ = timed(fwd)
print(f"forward {t_f:.2f} ms (no_grad {t_n:.2f}), forward+backward {t_b:.2f}, clamp loop {t_c:.2f}, Adam {t_o:.2f}; tensors {len(params)}")

# the fast step: flat arenas, fused clamp+Adam, manual two-lane backward (hint_amd.ConditionalFlowTrainer)
tr = hint_amd.ConditionalFlowTrainer(model)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l0, l1 = tr.step(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"ConditionalFlowTrainer: {B / dt:,.0f} samples/s, {dt * 1e3:.2f} ms/step, loss {float(l0) + float(l1):.4f}")
