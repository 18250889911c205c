"""Per-stage cycle breakdown of the forward kernel (diagnostic build with -DHINT_STAMPS).
   make -C hint_amd/csrc stamps && HINT_AMD_LIB=hint_amd/lib/libhint_amd_stamps.so python tools/stamps_fwd.py"""
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
x = torch.randn(B, d, device=dev)
buf = torch.zeros(8 * 512 + 64, dtype=torch.int64, device=dev)
assert lib.hint_debug_set_stamp_buffer(buf.data_ptr()) == 0, "not a stamps build"
with torch.no_grad():
    for _ in range(20):
        blk([x])
torch.cuda.synchronize()
s = buf.cpu()[:4096].view(8, 512)
names = {0: "start", 1: "x loaded+sync"}
for gi in range(4):
    for k, nm in enumerate(["build_v", "sync", "L1", "sync", "L2", "sync", "L3", "sync", "couple", "sync"]):
        names[2 + 12 * gi + k] = f"g{gi}:{nm}"
names[120] = "stored"
for gi in range(4):
    names[100 + 2 * gi] = f"g{gi}:L1 begun(L2)"; names[101 + 2 * gi] = f"g{gi}:L2 begun(L3)"; names[110 + gi] = f"g{gi}:L2 run done"
for k in range(60):
    names[60 + k] = "L2root:" + ["top", "decoded", "fetched", "aread", "mma"][k % 5] + f"#{k // 5}" if k < 15 else f"L2root:s{k}"
ids = sorted([i for i in names if s[0, i] != 0], key=lambda i: s[0, i].item())
t0 = s[:, 0].min().item()
print("stage".ljust(16) + "".join(f"w{w}".rjust(9) for w in range(8)) + "   (cycles since start; delta of wave 0)")
prev = None
for i in ids:
    row = [(s[w, i].item() - t0) for w in range(8)]
    dl = "" if prev is None else f"  +{row[0]-prev}"
    prev = row[0]
    print(names[i].ljust(16) + "".join(f"{v:9d}" for v in row) + dl)


# job starts inside the GEMM stages (ids 128 + (group*3 + stage)*12 + job), cycles after the wave's first job
print("\njob starts per wave (cycles since kernel start of the first job, then deltas):")
for gi in range(3):
    for st, nm in enumerate(['L1', 'L2', 'L3']):
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
