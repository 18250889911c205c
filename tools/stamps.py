"""Phase stamps of workgroup 0 of the forward and backward kernels inside a chained launch (diagnostic build):
   make -C hint_amd/csrc stamps && HINT_AMD_LIB=hint_amd/lib/libhint_amd_stamps.so python tools/stamps.py [workload]
Prints, per group of the tree, the cycles every wavefront spent in each phase (GEMM phases: busy time of
the wavefront; barriers: wait for the slowest one) for a block in the middle of the chain."""
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
    cfg["c_internal"] = [int(v) for v in os.environ["WIDTHS"].split(",")]     # (what-if runs: same tree, other hidden widths)
BLK = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # which block of the chain to print (512 stamp ids: block * groups < 32)
dev = torch.device("cuda:0")
lib = _lib.load()
lib.hint_debug_set_stamp_buffer.argtypes = [C.c_void_p]
torch.manual_seed(0)
flow = hint_amd.HintFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"], max_splits=int(os.environ.get("MAX_SPLITS", "-1"))).to(dev)
with torch.no_grad():
    for p in flow.parameters():
        p.data = 0.005 * torch.randn_like(p)
tr = hint_amd.FlowTrainer(flow, use_graph=False)
x = torch.randn(cfg["batch"], cfg["d"], device=dev)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
NW, IDS = 8, 512
buf = torch.zeros(NW * IDS, dtype=torch.int64, device=dev)
chain = tr._chain_for(x.shape[0])
B = x.shape[0]
z = torch.empty_like(x); J = torch.empty(B, device=dev); gx = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
n_groups = None


def show(title, st, per_block, labels, blocks):
    st = st.reshape(NW, IDS)
    nw = int((st[:, 0] != 0).sum())
    print(f"== {title}: {nw} wavefronts; cycles per phase, wave 0..{nw - 1} (block {blocks})")
    t_first = st[:nw][st[:nw] != 0].min(); t_last = st[:nw].max()
    print(f"   whole kernel (workgroup 0): {t_last - t_first} cycles")
    for cb in blocks:
        for gi in range(per_block):
            base = (cb * per_block + gi) * 16
            if title == "backward A" and (st[:nw, base + 5] != 0).all() and (st[:nw, base + 10] > st[:nw, base + 5]).all():
                # subtree phase (hint_sub.hpp): the stamps of the wavefront's last node
                for k, lab in enumerate(["  prefetch, unit records", "  coupling backward", "  two subnets", "  dW1 x 2, scatter", "  wait for the next tiles"]):
                    print(f"   blk {cb} grp {gi} {lab:26s} " + " ".join(f"{int(v):6d}" for v in st[:nw, base + 6 + k] - st[:nw, base + 5 + k]))
            row = st[:nw, base:base + 7]
            if title == "forward" and (st[:nw, base + 3] != 0).all() and (st[:nw, base + 8] != 0).all() and (st[:nw, base + 7] > st[:nw, base + 3]).all():
                # subtree phase (hint_sub.hpp): the stamps of the wavefront's last node
                for k, lab in enumerate(["  unit records", "  two subnets", "  tape stores", "  coupling", "  (to the level's end)"]):
                    print(f"   blk {cb} grp {gi} {lab:22s} " + " ".join(f"{int(v):6d}" for v in st[:nw, base + 4 + k] - st[:nw, base + 3 + k]))
            if (row == 0).all():
                continue
            for k, lab in enumerate(labels):
                d = row[:, k + 1] - row[:, k]
                if (row[:, k + 1] == 0).any() or (row[:, k] == 0).any():
                    continue
                print(f"   blk {cb} grp {gi} {lab:22s} " + " ".join(f"{int(v):6d}" for v in d))
            # inside the first row of the GEMM phase: thin layer, main chunks, last chunk, extra steps
            inner = st[:nw, base + 7:base + 16]
            for k, lab in enumerate(["  record loads", "  -> row0", "  row0 main chunks", "  row0 last chunk", "  row0 extra steps", "  other rows", "  deferred stores", "  next ring priming"]):
                if (inner[:, k] == 0).all() or (inner[:, k + 1] == 0).all():
                    continue
                d = np.where((inner[:, k] != 0) & (inner[:, k + 1] != 0), inner[:, k + 1] - inner[:, k], 0)
                print(f"   blk {cb} grp {gi} {lab:22s} " + " ".join(f"{int(v):6d}" for v in d))
            steps = st[:nw, 384 + ((cb * per_block + gi) & 7) * 16: 384 + ((cb * per_block + gi) & 7) * 16 + 16]
            if per_block <= 3 and (steps != 0).any():
                t0 = inner[:, 2]
                for k in range(16):
                    if (steps[:, k] == 0).all():
                        break
                    prev = t0 if k == 0 else steps[:, k - 1]
                    print(f"   blk {cb} grp {gi}     main step {k:2d}        " + " ".join(f"{int(v):6d}" for v in np.where(steps[:, k] != 0, steps[:, k] - prev, 0)))
        tb = st[:nw, cb * per_block * 16]
        te = st[:nw, (cb + 1) * per_block * 16] if (cb + 1) * per_block * 16 < IDS else None
        if te is not None and (te != 0).all():
            print(f"   blk {cb} total {int((te - tb).max())} cycles")
            tl = st[:nw, ((cb + 1) * per_block - 1) * 16 + 6]
            if (tl != 0).all():
                print(f"   blk {cb} -> {cb + 1} between the blocks (permutation, tape, thin vectors) " + " ".join(f"{int(v):6d}" for v in te - tl))


lib.hint_debug_set_stamp_buffer(buf.data_ptr())
_lib.check(lib.hint_chain_forward(chain, x.data_ptr(), None, z.data_ptr(), J.data_ptr(), None, None, stream), "fwd")
torch.cuda.synchronize()
fw = buf.cpu().numpy().copy()
buf.zero_()
_lib.check(lib.hint_chain_backward_parts(chain, x.data_ptr(), None, z.data_ptr(), None, gx.data_ptr(), None, 1.0 / B, -1.0 / B,
                                         1, 1, stream), "bwd")
torch.cuda.synchronize()
bw = buf.cpu().numpy().copy()
if os.environ.get("STAMPS_NPZ"):        # raw stamps for offline analysis: [wave][id]
    np.savez(os.environ["STAMPS_NPZ"], fw=fw.reshape(NW, IDS), bw=bw.reshape(NW, IDS))
lib.hint_debug_set_stamp_buffer(None)
stats = (C.c_int64 * 16)()
from hint_amd.hint import node_descs
nodes = flow.blocks[0].tree._flat_nodes()
descs, _, _, _ = node_descs(nodes)
lib.hint_plan_check(descs, len(nodes), cfg["d"], 0, 4.0, stats)
ng = int(stats[0])
print("groups per block:", ng)
if True:
    st = fw.reshape(NW, IDS)
    print("   (with subtree groups the general groups come first in the list below; the first slot behind them is the subtree phase and its barrier)")
    for rb in (256, 288):
        for k, lab in enumerate(["inputs, first operand", "k-loop", "epilogue", "fold + slab"]):
            print(f"   row {(rb - 256) // 32} {lab:24s}", " ".join(f"{int(v):6d}" for v in st[:, rb + k + 1] - st[:, rb + k]))
show("forward", fw, ng, ["P1 thin (VALU)", "barrier", "P2 rows (L2 L3)", "barrier", "P3 coupling", "barrier"], [BLK])
show("backward A", bw, ng + 1, ["Q1 couple/scatter", "barrier", "prefetch issue", "Q2 thin + barrier", "Q3 rows (g1 gv)", "commit+barrier"], [BLK])
