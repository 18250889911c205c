"""Dump the flat gradient of one backward pass (seeded flow and rows) to a .npy file, per parameter tensor compare two dumps:
   HINT_AMD_LIB=... python tools/grad_dump.py dump out.npy [d widths n_blocks B]     |     python tools/grad_dump.py cmp a.npy b.npy [d widths n_blocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hint_amd

mode = sys.argv[1]
d = int(sys.argv[4]) if len(sys.argv) > 4 else 6
widths = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [140, 70, 35, 17]
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 2
B = int(sys.argv[7]) if len(sys.argv) > 7 else 1015
torch.manual_seed(0)
flow = hint_amd.HintFlow(d, nb, widths)
with torch.no_grad():
    for p in flow.parameters():
        p.data = 0.06 * torch.randn(p.shape)
if mode == "dump":
    flow = flow.to("cuda:0")
    x = torch.randn(B, d).to("cuda:0")
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    tr._check_arenas(); tr.G.zero_(); tr._fwd_bwd(x, None)
    torch.cuda.synchronize()
    np.save(sys.argv[2], tr.G.cpu().numpy())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    flow = flow.to("cuda:0")
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False)
    for bi, ((lo, hi), eng) in enumerate(zip(tr.slices, tr.engines)):
        ga, gb = eng.split_flat(torch.from_numpy(a[lo:hi])), eng.split_flat(torch.from_numpy(b[lo:hi]))
        for p, x, y in zip(eng.params, ga, gb):
            name = [n for n, q in flow.blocks[bi].named_parameters() if q is p][0]
            err = float((x - y).norm() / max(float(y.norm()), 1e-30))
            if err > 1e-5:
                print(bi, name, tuple(x.shape), "rel diff %.3e" % err)
    print("compared")
