"""Throughput of the DROP-IN module path: the statements of /root/reference/train_unconditional.py:114-144 executed verbatim on
hint_amd's nn.Modules (autograd route, torch.optim.Adam, the per-parameter clamp), and where a step's time goes.
    python tools/module_path.py [workload] [steps] [--per-block] [--profile]
bench.py imports run() for its `extra.module_path` entry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(name="power_hint_8", steps=60, warmup=10, per_block=False, dev=None, profile=False):
    import bench, hint_amd
    from hint_amd import hint as H
    dev = dev or torch.device("cuda:0")
    cfg = bench.WORKLOADS[name]
    d, B = cfg["d"], cfg["batch"]
    torch.manual_seed(0)
    model = hint_amd.HintFlow(d, cfg["n_blocks"], cfg["c_internal"]).to(dev)
    if per_block and hasattr(model, "fuse_chain"):
        model.fuse_chain = False          # blocks and permutations called one by one, as FrEIA's ReversibleGraphNet would
    params_trainable = list(filter(lambda p: p.requires_grad, model.parameters()))
    for p in params_trainable:            # train_unconditional.py:165-167
        p.data = 0.005 * torch.randn_like(p.data)
    optim = torch.optim.Adam(params_trainable, lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
    x0 = torch.randn(B, d, device=dev)
    loss_history = []

    def body():                           # train_unconditional.py:114-144, one loader batch
        optim.zero_grad()
        batch_losses = []
        x = x0.clone()                    # (`x = x.to(c.device)`: the batch arrives as a fresh device tensor)
        x += 0.01 * torch.randn_like(x)
        z = model(x)
        log_jacobian = model.log_jacobian(x, run_forward=False)
        batch_losses.append(0.5 * torch.sum(z**2, dim=1).mean())
        batch_losses.append(-log_jacobian.mean())
        loss_total = sum(batch_losses)
        loss_history.append([l.item() for l in batch_losses])
        loss_total.backward()
        for p in params_trainable:
            p.grad.data.clamp_(-5.00, 5.00)
        optim.step()

    for _ in range(warmup):
        body()
    torch.cuda.synchronize()
    import gc
    gc.collect()                          # (the caller's garbage - bench.py's other workloads - is not this loop's)
    t0 = time.perf_counter()
    for _ in range(steps):
        body()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    # second pass: the same steps with hint_amd's entry points bracketed (host clock + HIP events on the launch stream)
    prof = H.profile_start() if hasattr(H, "profile_start") else None
    n2 = max(10, steps // 3)
    for _ in range(n2):
        body()
    torch.cuda.synchronize()
    split = H.profile_stop(prof, n2) if prof is not None else {}
    out = {"workload": f"{name}: d={d}, {cfg['n_blocks']} blocks, batch {B}", "route": "per block (FrEIA-style graph walk)" if per_block else "HintFlow",
           "samples_per_sec": B / wall, "ms_per_step": wall * 1e3, "steps": steps, "last_losses": loss_history[-1], **split}
    if profile:
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20):
            body()
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
        out["cprofile"] = s.getvalue()
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    r = run(args[0] if args else "power_hint_8", int(args[1]) if len(args) > 1 else 60, per_block="--per-block" in sys.argv,
            profile="--profile" in sys.argv)
    cp = r.pop("cprofile", None)
    import json
    print(json.dumps(r))
    if cp:
        print(cp)
