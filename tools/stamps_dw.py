"""Phase stamps of EVERY wavefront of part B (hint_wgrad_kernel) inside a chained launch (diagnostic build):
   make -C hint_amd/csrc stamps && HINT_AMD_LIB=hint_amd/lib/libhint_amd_stamps.so python tools/stamps_dw.py [workload]
Prints the phases' durations (median / mean over the wavefronts, shader-clock ticks of s_memtime), the lifetime of a
workgroup, and how many workgroups were inside their row loops over the launch (a timeline of 20 bins)."""
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
dev = torch.device("cuda:0")
lib = _lib.load()
lib.hint_debug_set_dw_stamp_buffer.argtypes = [C.c_void_p]
torch.manual_seed(0)
flow = hint_amd.HintFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"]).to(dev)
with torch.no_grad():
    for p in flow.parameters():
        p.data = 0.005 * torch.randn_like(p)
tr = hint_amd.FlowTrainer(flow, use_graph=False)
x = torch.randn(cfg["batch"], cfg["d"], device=dev)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
NWG, NW, IDS = 1 << 16, int(os.environ.get("DW_WAVES", "8")), 8
buf = torch.zeros(NWG * NW * IDS, dtype=torch.int64, device=dev)
chain = tr._chain_for(x.shape[0])
B = x.shape[0]
z = torch.empty_like(x); gx = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
# a forward + part A so that tape and workspace hold a real step, then part B alone with the stamps on
tr._fwd_bwd(x, None)
torch.cuda.synchronize()
lib.hint_debug_set_dw_stamp_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.hint_chain_backward_parts(chain, x.data_ptr(), None, z.data_ptr(), None, gx.data_ptr(), None, 1.0 / B, -1.0 / B, 1, 2, stream), "part B")
e1.record()
torch.cuda.synchronize()
lib.hint_debug_set_dw_stamp_buffer(None)
us = e0.elapsed_time(e1) * 1e3
st = buf.cpu().numpy().reshape(NWG, NW, IDS)
used = st[:, :, 0] != 0
nwg = int(used.any(axis=1).sum())
w = st[used]                       # [waves, ids]
w = w[(w[:, 5] != 0) & (w[:, 1] != 0)]
t0 = int(w[:, 0].min())
span = int(w[:, 5].max()) - t0
print(f"{name}: part B + reduction {us:.1f} us between events; {nwg} workgroups, {len(w)} wavefronts with work; "
      f"first stamp to last {span} ticks ({span / us:.1f} ticks per us)")
labels = ["records arrive (block pointers, job)", "row loop (incl. its parameter / first-input latency)", "partials -> LDS",
          "wait for the workgroup's wavefronts", "reduce + stores (to the last store's retirement)"]
for k, lab in enumerate(labels):
    dlt = (w[:, k + 1] - w[:, k]).astype(np.float64)
    ok = (w[:, k + 1] != 0) & (w[:, k] != 0)
    if ok.any():
        print(f"   {lab:58s} median {np.median(dlt[ok]):8.0f}  mean {dlt[ok].mean():8.0f}  max {dlt[ok].max():8.0f}")
life = (w[:, 5] - w[:, 0]).astype(np.float64)
print(f"   wavefront lifetime                                         median {np.median(life):8.0f}  mean {life.mean():8.0f}  max {life.max():8.0f}")
# timeline: wavefronts inside their row loop / alive, per bin
bins = 20
edges = t0 + np.arange(bins + 1) * (span / bins)
alive = [(int(((w[:, 0] < b1) & (w[:, 5] > b0)).sum()), int(((w[:, 1] < b1) & (w[:, 2] > b0)).sum())) for b0, b1 in zip(edges[:-1], edges[1:])]
print("   wavefronts alive / in the row loop per 1/20 of the launch:")
print("     " + " ".join(f"{a:5d}" for a, _ in alive))
print("     " + " ".join(f"{r:5d}" for _, r in alive))
