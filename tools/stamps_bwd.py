"""Per-stage cycle breakdown of the backward kernel, part A (diagnostic build with -DHINT_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hint_amd
from hint_amd import _lib
d, widths, B = 6, [140, 70, 35, 17], 4096
if len(sys.argv) > 1:
    d = int(sys.argv[1]); widths = [int(v) for v in sys.argv[2].split(",")]; B = int(sys.argv[3])
dev = "cuda:0"
lib = _lib.load()
blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=widths).to(dev)
x = torch.randn(B, d, device=dev, requires_grad=True)
buf = torch.zeros(8 * 512 + 64, dtype=torch.int64, device=dev)
(z,) = blk([x]); J = blk.jacobian(None)
L = (0.5 * (z ** 2).sum(1) - J).mean()
for _ in range(5):
    L.backward(retain_graph=True)
torch.cuda.synchronize()
assert lib.hint_debug_set_stamp_buffer(buf.data_ptr()) == 0, "not a stamps build"
lib.hint_debug_set_backward_stages(1)
L.backward(retain_graph=True)
torch.cuda.synchronize()
lib.hint_debug_set_backward_stages(3)
s = buf.cpu()[:4096].view(8, 512)
names = {0: "start", 1: "loaded+sync", 120: "stored"}
# phase stamps of the tape-based backward kernel (ids 2 + 20*group + k)
stages = ["-", "-", "-", "-", "commit s,a2", "sync", "-", "-", "couple", "sync",
          "-", "-", "g2+dW3 tiles", "sync", "g1(+copy g2,colsum)", "sync", "dv+dW1 tiles", "sync", "scatter+build v", "sync"]
for gi in range(5):
    for k, nm in enumerate(stages):
        names[2 + 20 * gi + k] = f"g{gi}:{nm}"
for gi in range(4):
    names[104 + 4 * gi] = f"g{gi}:  o3 ojobs done"; names[105 + 4 * gi] = f"g{gi}:  g2 copied out"
    names[106 + 4 * gi] = f"g{gi}:  colsum(g2) done"; names[107 + 4 * gi] = f"g{gi}:  dv stage done"
ids = sorted([i for i in names if s[0, i] != 0], key=lambda i: s[0, i].item())
t0 = s[:, 0].min().item()
prev = None
for i in ids:
    row = [(s[w, i].item() - t0) for w in range(8)]
    dl = "" if prev is None else f"  +{row[0]-prev}"
    prev = row[0]
    print(names[i].ljust(28) + f"{row[0]:9d} {max(row):9d}" + dl)

# job starts inside the GEMM stages (ids 128 + (group*3 + stage)*12 + job), cycles after the wave's first job
print("\njob starts per wave (cycles since kernel start of the first job, then deltas):")
for gi in range(3):
    for st, nm in enumerate(['g2+dW3', 'g1', 'dv+dW1']):
        base = 128 + (gi * 3 + st) * 12
        if s[:, base].max().item() == 0: continue
        print(f"g{gi}:{nm}")
        for w in range(8):
            ts = [s[w, base + k].item() for k in range(12) if s[w, base + k].item() != 0]
            if ts: print(f"   w{w}  start {ts[0]-t0:7d}  " + " ".join(f"+{b-a}" for a, b in zip(ts, ts[1:])))

# sequential section log of one stage (-DHINT_STAMP_STAGE=<job base id>): section id, cycles since the previous entry
if s[:, 256].max().item() != 0:
    print("\nsection log (K:+cycles)  1 job top, 3 run_job entered, 5 k-loop issued, 6 next weights requested, 7 job done, 8 outer operands read, 9 outer: next weights requested")
    for w in range(8):
        ent = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in s[w, 256:512].tolist() if v != 0]
        if ent: print(f"   w{w} @{ent[0][1]-t0}: " + " ".join(f"{k}:+{b-a}" for (k, b), (_, a) in zip(ent[1:], ent[:-1])))
