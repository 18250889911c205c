#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (`/root/reference/hint.py`).

Runs only in the dev container (the reference does not exist on the GPU box).  FrEIA is not
installed, so `FrEIA.modules.orthogonal.HouseholderPerm` is a stand-in (a fixed random orthogonal
matrix, stored in the fixture); it is only touched by the two reshuffle=True cases
(hint.py:36-39): those pin where the reference applies the per-node matrices, not FrEIA's
Householder construction.  Nothing from the reference is
copied: the fixtures hold inputs' checksums and the reference's numerical OUTPUTS.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import (BLOCK_CASES, CHAIN_CASES, REV_GRAD_CASES, ROOT, checksum, make_block_inputs,  # noqa: E402
                   make_chain_inputs, norm_case)

sys.path.insert(0, ROOT)
from oracle import hint_oracle as orc  # noqa: E402  (only for the key/shape contract check)

REF = "/root/reference"


def import_reference():
    for name in ("FrEIA", "FrEIA.modules", "FrEIA.modules.orthogonal"):
        sys.modules.setdefault(name, types.ModuleType(name))

    class HouseholderPerm:
        """stand-in with FrEIA's call protocol (hint.py:37,65,94): module([x], rev) -> [x W] / [x W^T]"""
        rs = np.random.RandomState(0)

        def __init__(self, dims_in, dims_c=[], n_reflections=1, fixed=True):
            D = dims_in[0][0]
            q, r = np.linalg.qr(HouseholderPerm.rs.standard_normal((D, D)))
            self.W = torch.from_numpy(np.ascontiguousarray((q * np.sign(np.diag(r))).astype(np.float32)))

        def __call__(self, x, c=[], rev=False):
            return [x[0] @ (self.W.t() if rev else self.W)]

    sys.modules["FrEIA.modules.orthogonal"].HouseholderPerm = HouseholderPerm
    sys.path.insert(0, REF)
    import hint as ref_hint
    return ref_hint


def build_ref_block(ref_hint, c):
    blk = ref_hint.HierarchicalAffineCouplingBlock(
        [(c["d"],)], dims_c=list(c["dims_c"]), c_internal=list(c["c_internal"]),
        clamp=c["clamp"], max_splits=c["max_splits"], min_split_size=c["min_split_size"],
        reshuffle=c["reshuffle"])
    return blk


def check_contract(blk, c):
    nodes = orc.build_nodes(c["d"], c["dims_c"], c["c_internal"], c["max_splits"], c["min_split_size"])
    shapes = orc.param_shapes(nodes)
    sd = blk.state_dict()
    assert list(sd.keys()) == list(shapes.keys()), "state_dict key order differs from reference"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    return shapes


def gen_block(ref_hint, case):
    c = norm_case(case)
    torch.manual_seed(0)
    sys.modules["FrEIA.modules.orthogonal"].HouseholderPerm.rs = np.random.RandomState(len(case["name"]) + 7 * case["d"])
    blk = build_ref_block(ref_hint, c)
    shapes = check_contract(blk, c)
    params, x_np, conds_np = make_block_inputs(case, shapes)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})

    x = torch.from_numpy(x_np).requires_grad_(True)
    conds = [torch.from_numpy(a).requires_grad_(True) for a in conds_np]
    (z,) = blk.forward([x], c=conds, rev=False)
    J = blk.jacobian(None)
    L = (0.5 * torch.sum(z ** 2, dim=1) - J).mean()
    L.backward()
    out = dict(z=z.detach().numpy(), J=J.detach().numpy(), L=np.float64(L.item()),
               gx=x.grad.numpy())
    for i, cc in enumerate(conds):
        out[f"gc{i}"] = cc.grad.numpy()
    named = dict(blk.named_parameters())
    for k in shapes:
        out["g:" + k] = named[k].grad.numpy()
    with torch.no_grad():
        (xr,) = blk.forward([z.detach()], c=[cc.detach() for cc in conds], rev=True)
        Jr = blk.jacobian(None)
        # inverse applied to fresh data (sampling direction), not only the round trip
        (xs,) = blk.forward([x.detach()], c=[cc.detach() for cc in conds], rev=True)
        Js = blk.jacobian(None)
    out.update(x_rec=xr.numpy(), J_rev=Jr.numpy(), x_inv=xs.numpy(), J_inv=Js.numpy())
    out["in_checksum"] = np.float64(checksum(list(params.values()) + [x_np] + conds_np))
    out["keys"] = np.array(list(shapes.keys()))
    if c["reshuffle"]:            # the stand-in's matrices, by module path ("tree", "tree.upper", ...)
        for path, m in blk.named_modules():
            if getattr(m, "perm", None) is not None:
                out["perm:" + path] = m.perm.W.numpy()
    return out


def gen_block_rev(ref_hint, case):
    """gradients through the inverse direction (hint.py:82-88 under autograd), same parameters and inputs as gen_block"""
    c = norm_case(case)
    torch.manual_seed(0)
    sys.modules["FrEIA.modules.orthogonal"].HouseholderPerm.rs = np.random.RandomState(len(case["name"]) + 7 * case["d"])
    blk = build_ref_block(ref_hint, c)
    shapes = check_contract(blk, c)
    params, x_np, conds_np = make_block_inputs(case, shapes)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    z = torch.from_numpy(x_np).requires_grad_(True)
    conds = [torch.from_numpy(a).requires_grad_(True) for a in conds_np]
    (x,) = blk.forward([z], c=conds, rev=True)
    J = blk.jacobian(None)
    L = (0.5 * torch.sum(x ** 2, dim=1) - J).mean()
    L.backward()
    out = dict(x_inv=x.detach().numpy(), J_inv=J.detach().numpy(), L=np.float64(L.item()), gz=z.grad.numpy())
    for i, cc in enumerate(conds):
        out[f"gc{i}"] = cc.grad.numpy()
    named = dict(blk.named_parameters())
    for k in shapes:
        out["g:" + k] = named[k].grad.numpy()
    out["in_checksum"] = np.float64(checksum(list(params.values()) + [x_np] + conds_np))
    return out


def gen_chain(ref_hint, case):
    c = norm_case(dict(case, dc=0))
    torch.manual_seed(0)
    blocks = [build_ref_block(ref_hint, c) for _ in range(case["n_blocks"])]
    shapes = check_contract(blocks[0], c)
    params, perms, xs = make_chain_inputs(case, shapes)
    for blk, P in zip(blocks, params):
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    Ws = [None if p is None else torch.from_numpy(p) for p in perms]

    def fwd(x):
        J = 0
        for blk, W in zip(blocks, Ws):
            if W is not None:
                x = x @ W
            (x,) = blk.forward([x])
            J = J + blk.jacobian(None)
        return x, J

    out = {}
    all_params = [p for blk in blocks for p in blk.parameters()]
    if case["steps"] == 0:
        with torch.no_grad():
            z, J = fwd(torch.from_numpy(xs[0]))
        l0 = 0.5 * torch.sum(z ** 2, dim=1).mean()
        l1 = -J.mean()
        out.update(z=z.numpy(), J=J.numpy(), l0=np.float64(l0.item()), l1=np.float64(l1.item()),
                   nll=np.float64(l0.item() + l1.item() + 0.5 * case["d"] * np.log(2 * np.pi)))
    else:
        # train_unconditional.py:120-144,174-176 with the first-epochs lr (:191-193)
        opt = torch.optim.Adam(all_params, lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4,
                               weight_decay=1.86e-5)
        losses = []
        for x_np in xs:
            opt.zero_grad()
            z, J = fwd(torch.from_numpy(x_np))
            l0 = 0.5 * torch.sum(z ** 2, dim=1).mean()
            l1 = -J.mean()
            (l0 + l1).backward()
            for p in all_params:
                p.grad.data.clamp_(-5.0, 5.0)
            opt.step()
            losses.append([l0.item(), l1.item()])
        out["losses"] = np.array(losses, dtype=np.float64)
        for bi, blk in enumerate(blocks):
            for k, v in blk.state_dict().items():
                out[f"final:{bi}:{k}"] = v.numpy()
    for i, W in enumerate(perms):      # QR is LAPACK-dependent: store the matrices themselves
        if W is not None:
            out[f"perm:{i}"] = W
    out["in_checksum"] = np.float64(checksum([v for P in params for v in P.values()] + xs))
    return out


def main():
    ref_hint = import_reference()
    for case in BLOCK_CASES:
        if case["name"] in REV_GRAD_CASES:
            out = gen_block_rev(ref_hint, case)
            np.savez_compressed(os.path.join(HERE, f"revgrad_{case['name']}.npz"), **out)
            print("revgrad", case["name"], "L=%.6f" % out["L"])
    if len(sys.argv) > 1 and sys.argv[1] == "rev":          # only the fixtures above
        return
    for case in BLOCK_CASES:
        out = gen_block(ref_hint, case)
        np.savez_compressed(os.path.join(HERE, f"block_{case['name']}.npz"), **out)
        print("block", case["name"], "L=%.6f" % out["L"], "max|J|=%.3f" % np.abs(out["J"]).max())
    for case in CHAIN_CASES:
        out = gen_chain(ref_hint, case)
        np.savez_compressed(os.path.join(HERE, f"chain_{case['name']}.npz"), **out)
        print("chain", case["name"])


if __name__ == "__main__":
    main()
