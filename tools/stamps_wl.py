"""Phase stamps of workgroup 0 of the wave-local kernels (hint_wl_fwd.hip / hint_wl_bwd.hip) inside a chained launch
(diagnostic build):
   make -C hint_amd/csrc stamps && HINT_AMD_LIB=hint_amd/lib/libhint_amd_stamps.so python tools/stamps_wl.py [workload] [block]
Prints, per tree level of one block in the middle of the chain, the cycles every wavefront spent in its rows, waiting at
the barrier and in the element-wise work."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hint_amd
from hint_amd import _lib
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "power_hint_8"
cfg = dict(WORKLOADS[name])
if os.environ.get("WIDTHS"):
    cfg["c_internal"] = [int(v) for v in os.environ["WIDTHS"].split(",")]
BLK = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
lib = _lib.load()
lib.hint_debug_set_stamp_buffer.argtypes = [C.c_void_p]
torch.manual_seed(0)
flow = hint_amd.HintFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"]).to(dev)
with torch.no_grad():
    for p in flow.parameters():
        p.data = 0.005 * torch.randn_like(p)
tr = hint_amd.FlowTrainer(flow, use_graph=False)
B = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["batch"]
x = torch.randn(B, cfg["d"], device=dev)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
NW, IDS = 8, 512
buf = torch.zeros(NW * IDS, dtype=torch.int64, device=dev)
chain = tr._chain_for(B)
z = torch.empty_like(x); J = torch.empty(B, device=dev); gx = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
from hint_amd.hint import node_descs
nodes = flow.blocks[0].tree._flat_nodes()
descs, _, _, _ = node_descs(nodes)
stats = (C.c_int64 * 16)()
lib.hint_plan_check(descs, len(nodes), cfg["d"], 0, 4.0, stats)
ng = int(stats[0])


def show(title, st, per_block, labels, rows=True):
    st = st.reshape(NW, IDS)
    nw = int((st[:, 0] != 0).sum())
    nz = st[:nw][st[:nw] != 0]
    print(f"== {title}: {nw} wavefronts, whole kernel (workgroup 0) {int(nz.max() - nz.min())} cycles")
    for gi in range(per_block):
        base = (BLK * per_block + gi) * 8
        row = st[:nw, base:base + len(labels) + 1]
        for k, lab in enumerate(labels):
            if (row[:, k + 1] == 0).any() or (row[:, k] == 0).any():
                continue
            d = row[:, k + 1] - row[:, k]
            print(f"   blk {BLK} grp {gi} {lab:24s} " + " ".join(f"{int(v):6d}" for v in d))
    if rows:
        for slot in range(4):
            for r in range(2):
                base = 256 + slot * 64 + r * 32
                row = st[:nw, base:base + 5]
                if (row == 0).all():
                    continue
                for k, lab in enumerate(["setup (inputs, B frag 0)", "k-loop", "epilogue", "flush (fold, slab)"]):
                    d = np.where((row[:, k] != 0) & (row[:, k + 1] != 0), row[:, k + 1] - row[:, k], 0)
                    print(f"   grp-slot {slot} row {r} {lab:26s} " + " ".join(f"{int(v):6d}" for v in d))
                steps = st[:nw, base + 8: base + 24]
                prev = row[:, 1]
                for k in range(16):
                    if (steps[:, k] == 0).all():
                        break
                    print(f"   grp-slot {slot} row {r}   step {k:2d}                  " + " ".join(f"{int(v):6d}" for v in np.where(steps[:, k] != 0, steps[:, k] - prev, 0)))
                    prev = steps[:, k]
    tb = st[:nw, BLK * per_block * 8]
    te = st[:nw, (BLK + 1) * per_block * 8]
    if (te != 0).all():
        print(f"   blk {BLK} total {int((te - tb).max())} cycles")


lib.hint_debug_set_stamp_buffer(buf.data_ptr())
_lib.check(lib.hint_chain_forward(chain, x.data_ptr(), None, z.data_ptr(), J.data_ptr(), None, None, stream), "fwd")
torch.cuda.synchronize()
fw = buf.cpu().numpy().copy()
buf.zero_()
_lib.check(lib.hint_chain_backward_parts(chain, x.data_ptr(), None, z.data_ptr(), None, gx.data_ptr(), None, 1.0 / B, -1.0 / B,
                                         1, 1, stream), "bwd")
torch.cuda.synchronize()
bw = buf.cpu().numpy().copy()
buf.zero_()
_lib.check(lib.hint_chain_inverse(chain, z.data_ptr(), None, x.data_ptr(), J.data_ptr(), None, stream), "inv")
torch.cuda.synchronize()
iv = buf.cpu().numpy().copy()
lib.hint_debug_set_stamp_buffer(None)
print("groups per block:", ng)
show("forward (training)", fw, ng, ["rows", "param commit", "barrier wait", "coupling", "tape level store"])
show("inverse", iv, ng, ["rows", "param commit", "barrier wait", "coupling", "tape level store"])
show("backward A", bw, ng + 1, ["scatter + coupling bwd", "rows", "commits", "barrier wait"])
