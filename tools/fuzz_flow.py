"""Randomised consistency sweep of the flow-level paths (GPU): the chained launches of FlowTrainer (one
forward / two backward launches for the whole flow, fused permutations, in-kernel loss) against the
module-by-module autograd path of HintFlow, on random flows (lanes, widths, blocks, condition,
node permutations, batch sizes incl. ragged and > one row tile per workgroup).
   python tools/fuzz_flow.py [n_cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hint_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
worst = dict(loss=0.0, grad=0.0)
for case in range(n_cases):
    d = rng.choice([1, 2, 3, 5, 6, 8, 9, 13, 21, 43, 64, 100])
    widths = [rng.choice([8, 16, 17, 24, 33, 64, 70, 140]) for _ in range(rng.randint(1, 3))]
    nb = rng.randint(1, 4)
    dc = rng.choice([0, 0, 2, 4])
    B = rng.choice([1, 16, 17, 100, 257, 1000, 4113])
    perm_first = rng.random() < 0.5
    reshuffle = rng.random() < 0.25
    torch.manual_seed(case)
    flow = hint_amd.HintFlow(d, nb, widths, ndim_c=dc, perm_first=perm_first, reshuffle=reshuffle).to(dev)
    for p in flow.parameters():
        p.data.mul_(0.3 if d >= 43 else 0.6)           # (default init makes long chains of wide blocks explode)
    x = torch.randn(B, d, device=dev)
    c = torch.randn(B, dc, device=dev) if dc else None
    z = flow(x, c=c) if dc else flow(x)
    J = flow.log_jacobian(run_forward=False)
    l0, l1 = 0.5 * (z ** 2).sum(1).mean(), -J.mean()
    (l0 + l1).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in flow.parameters()]).clone()
    res = {}
    for chain in (True, False):
        tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False, use_chain=chain)
        tr._check_arenas(); tr.G.zero_(); tr._fwd_bwd(x, c); torch.cuda.synchronize()
        fast = torch.cat([g.reshape(-1) for e, (a, b) in zip(tr.engines, tr.slices) for g in e.split_flat(tr.G[a:b])])
        s = tr.loss_acc.sum(0)
        res[chain] = (float(s[0]) / B, -float(s[1]) / B, fast, tr._chainable)
    for chain, (a0, a1, fast, chained) in res.items():
        el = max(abs(a0 - float(l0)) / max(1.0, abs(float(l0))), abs(a1 - float(l1)) / max(1.0, abs(float(l1))))
        eg = ((fast - ref).abs().max() / (ref.abs().max() + 1e-30)).item()
        worst["loss"] = max(worst["loss"], el); worst["grad"] = max(worst["grad"], eg)
        bad = el > 1e-5 or eg > 1e-4
        if bad and el <= 1e-5 and eg < 2e-3:
            # The module path multiplies the fixed permutations with torch (rocBLAS), the chained launches in-kernel: values that
            # differ in the last bit can put a hidden unit of one row on the other side of its ReLU kink.  That shows as ONE unit's
            # layer tensors off by about a row's share and everything else at 1e-5: accepted, and said.
            sizes = [p.numel() for p in flow.parameters()]
            names = [n for n, _ in flow.named_parameters()]
            off, pos = [], 0
            for n_, k in zip(names, sizes):
                e1 = ((fast[pos:pos + k] - ref[pos:pos + k]).abs().max() / (ref.abs().max() + 1e-30)).item()
                if e1 > 2e-5:
                    off.append(n_)
                pos += k
            units = {n_.rsplit(".", 2)[0] for n_ in off}           # (.../tree.lower.s: a subnet)
            if len(units) <= 3:
                bad = False
                print(f"kink case {case}: d={d} widths={widths} blocks={nb} B={B}: grad {eg:.1e} in {sorted(units)}", flush=True)
        if bad or (case % 10 == 0 and chain):
            print(("BAD " if bad else "ok  ") + f"case {case}: d={d} widths={widths} blocks={nb} dc={dc} B={B} perm_first={perm_first} "
                  f"reshuffle={reshuffle} chained={chained and chain}: loss {el:.1e} grad {eg:.1e}", flush=True)
        assert not bad
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()})
