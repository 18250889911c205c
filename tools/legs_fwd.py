import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hint_amd
from hint_amd import _lib
d, widths, B = 6, [140, 70, 35, 17], 4096
dev = torch.device("cuda:0")
lib = _lib.load()
blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=widths).to(dev)
x = torch.randn(B, d, device=dev)
with torch.no_grad():
    blk([x])
eng = blk.tree._engine
z = torch.empty_like(x); J = torch.empty(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
def f(): lib.hint_block_forward(eng.plan, eng.arena.data_ptr(), eng.packed.data_ptr(), x.data_ptr(), None, z.data_ptr(), J.data_ptr(), None, B, st)
for _ in range(20): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): f()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("HINT_AMD_LIB", "default").split("/")[-1], "fwd us:", round(e0.elapsed_time(e1) * 1e3 / 200, 2))
