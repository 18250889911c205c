"""Shared helpers for the test-suite: fixture loading and oracle plumbing."""
import os

import numpy as np
import torch

from cases import (BLOCK_CASES, CHAIN_CASES, REV_GRAD_CASES, checksum, make_block_inputs,  # noqa: F401
                   make_chain_inputs, norm_case)
from oracle import hint_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_block_case(case):
    """-> (normalised case, nodes, shapes, params(np), x(np), conds(np list), golden npz)"""
    c = norm_case(case)
    nodes = orc.build_nodes(c["d"], c["dims_c"], c["c_internal"], c["max_splits"], c["min_split_size"])
    shapes = orc.param_shapes(nodes)
    params, x, conds = make_block_inputs(case, shapes)
    g = np.load(os.path.join(GOLDEN, f"block_{c['name']}.npz"))
    assert abs(checksum(list(params.values()) + [x] + conds) - float(g["in_checksum"])) < 1e-6, \
        "regenerated inputs differ from the ones the golden vectors were computed on"
    assert list(g["keys"]) == list(shapes.keys())
    return c, nodes, shapes, params, x, conds, g


def load_revgrad(c, g):
    """golden gradients through the inverse of an already loaded block case (None if the case has none)"""
    if c["name"] not in REV_GRAD_CASES:
        return None
    r = np.load(os.path.join(GOLDEN, f"revgrad_{c['name']}.npz"))
    assert abs(float(r["in_checksum"]) - float(g["in_checksum"])) < 1e-6
    return r


def load_chain_case(case):
    c = norm_case(dict(case, dc=0))
    nodes = orc.build_nodes(c["d"], c["dims_c"], c["c_internal"], c["max_splits"], c["min_split_size"])
    shapes = orc.param_shapes(nodes)
    params, perms, xs = make_chain_inputs(case, shapes)
    g = np.load(os.path.join(GOLDEN, f"chain_{case['name']}.npz"))
    assert abs(checksum([v for P in params for v in P.values()] + xs) - float(g["in_checksum"])) < 1e-6
    perms = [None if p is None else g[f"perm:{i}"] for i, p in enumerate(perms)]
    return c, nodes, shapes, params, perms, xs, g


def case_perms(g):
    """{node path: matrix} of a reshuffle=True fixture (empty for the others)"""
    return {k[len("perm:"):]: g[k] for k in g.files if k.startswith("perm:")}


def to_torch(P, dtype=torch.float32):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in P.items()}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
