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
buf = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
assert lib.hint_debug_set_stamp_buffer(buf.data_ptr()) == 0, "not a stamps build"
with torch.no_grad():
    for _ in range(20):
        blk([x])
torch.cuda.synchronize()
s = buf.cpu().view(8, 128)
names = {0: "start", 1: "x loaded+sync"}
for gi in range(4):
    for k, nm in enumerate(["build_v", "sync", "L1", "sync", "L2", "sync", "L3", "sync", "couple", "sync"]):
        names[2 + 12 * gi + k] = f"g{gi}:{nm}"
names[120] = "stored"
for k in range(60):
    names[60 + k] = "L2root:" + ["top", "decoded", "fetched", "aread", "mma"][k % 5] + f"#{k // 5}" if k < 15 else f"L2root:s{k}"
ids = [i for i in sorted(names) if s[0, i] != 0]
t0 = s[:, 0].min().item()
print("stage".ljust(16) + "".join(f"w{w}".rjust(9) for w in range(8)) + "   (cycles since start; delta of wave 0)")
prev = None
for i in ids:
    row = [(s[w, i].item() - t0) for w in range(8)]
    dl = "" if prev is None else f"  +{row[0]-prev}"
    prev = row[0]
    print(names[i].ljust(16) + "".join(f"{v:9d}" for v in row) + dl)
