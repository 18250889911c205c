"""Golden-vector case list and deterministic input generation (shared by make_golden.py,
which runs only in the dev container where /root/reference exists, and by the tests).

Inputs and parameters are drawn from numpy's legacy `RandomState` (bit-stable across
platforms and numpy versions), so the committed fixtures only have to hold the reference's
OUTPUTS plus a checksum of the inputs they were computed from.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name, d, dims_c, c_internal, clamp, max_splits, min_split_size, init ('randn'|'uniform'), B
BLOCK_CASES = [
    dict(name="power8_randn",   d=6,   dc=0, c_internal=[140, 70, 35, 17], init="randn"),
    dict(name="power8_unif",    d=6,   dc=0, c_internal=[140, 70, 35, 17], init="uniform"),
    dict(name="power4_unif",    d=6,   dc=0, c_internal=[200, 100, 50, 25], init="uniform"),
    dict(name="gas8_randn",     d=8,   dc=0, c_internal=[128, 64, 32, 16], init="randn"),
    dict(name="gas8_unif",      d=8,   dc=0, c_internal=[128, 64, 32, 16], init="uniform"),
    dict(name="mini43_unif",    d=43,  dc=0, c_internal=[67, 33, 16, 8], init="uniform"),
    dict(name="mini42_unif",    d=42,  dc=0, c_internal=[67, 33, 16, 8], init="uniform"),
    dict(name="plus100_unif",   d=100, dc=0, c_internal=[56, 28, 14], init="uniform", B=32),
    dict(name="plus100_c4",     d=100, dc=4, c_internal=[32, 16, 8], init="uniform", B=32),
    dict(name="cond_d6_c4",     d=6,   dc=4, c_internal=[24, 12], init="uniform"),
    dict(name="cond_two_c",     d=7,   dc=(2, 3), c_internal=[20], init="uniform"),
    dict(name="splits0_d8",     d=8,   dc=0, c_internal=[32, 16, 8], init="uniform", max_splits=0),
    dict(name="splits1_d8",     d=8,   dc=0, c_internal=[32, 16, 8], init="uniform", max_splits=1),
    dict(name="splits2_d8",     d=8,   dc=0, c_internal=[32, 16, 8], init="uniform", max_splits=2),
    dict(name="minsplit3_d12",  d=12,  dc=0, c_internal=[24, 12, 6], init="uniform", min_split_size=3),
    dict(name="cint_empty_d6",  d=6,   dc=0, c_internal=[], init="uniform"),
    dict(name="cint_one_d9",    d=9,   dc=0, c_internal=[19], init="uniform"),
    dict(name="clamp2_d6",      d=6,   dc=0, c_internal=[30, 15], init="uniform", clamp=2.0),
    dict(name="odd_d5",         d=5,   dc=0, c_internal=[17, 9], init="uniform"),
    dict(name="tiny_d2",        d=2,   dc=0, c_internal=[8], init="uniform"),
    dict(name="tiny_d3",        d=3,   dc=0, c_internal=[8], init="uniform"),
    dict(name="minsplit1_d4",   d=4,   dc=0, c_internal=[8, 4, 4], init="uniform", min_split_size=1),
    dict(name="big_s_d6",       d=6,   dc=0, c_internal=[40, 20], init="uniform", wscale=6.0),
    # reshuffle=True (hint.py:36-39,64-65,93-94) with a stand-in for FrEIA's HouseholderPerm: pins WHERE
    # the per-node orthogonal matrices act, not how FrEIA builds them; the matrices are in the fixture
    dict(name="reshuffle_d6",   d=6,   dc=0, c_internal=[24, 12], init="uniform", reshuffle=True),
    dict(name="reshuffle_d9c2", d=9,   dc=2, c_internal=[20, 10, 6], init="uniform", reshuffle=True),
]

# block cases that also pin the gradients THROUGH the inverse (rev=True is differentiable in the reference: hint.py:82-88):
# revgrad_<name>.npz = d/dz, d/dc, d/dW of mean(0.5*|x|^2 - J) with (x, J) = block(z, rev=True), z = the case's input
REV_GRAD_CASES = ["power8_unif", "gas8_unif", "mini43_unif", "plus100_c4", "cond_d6_c4", "cond_two_c",
                  "splits1_d8", "minsplit3_d12", "cint_empty_d6", "clamp2_d6", "odd_d5", "tiny_d2",
                  "reshuffle_d6", "reshuffle_d9c2"]

# chained flows: (blocks chained by stored orthogonal matrices) + K Adam steps
CHAIN_CASES = [
    dict(name="chain_power4", d=6, n_blocks=4, c_internal=[200, 100, 50, 25], B=64, steps=0),
    dict(name="chain_train5", d=6, n_blocks=3, c_internal=[32, 16], B=128, steps=5),
]


def norm_case(case):
    c = dict(case)
    c.setdefault("clamp", 4.0)
    c.setdefault("max_splits", -1)
    c.setdefault("min_split_size", 2)
    c.setdefault("B", 64)
    c.setdefault("wscale", 1.0)
    c.setdefault("reshuffle", False)
    dc = c.get("dc", 0)
    if isinstance(dc, int):
        c["dims_c"] = [(dc,)] if dc > 0 else []
    else:
        c["dims_c"] = [(int(v),) for v in dc]
    return c


def _seed_of(name: str) -> int:
    return int.from_bytes(name.encode(), "little") % (2 ** 31 - 1)


def draw_params(shapes: dict, init: str, rs: np.random.RandomState, wscale: float = 1.0):
    """shapes: ordered {state_dict key: shape}.  'randn' = 0.005*N(0,1)
    (train_unconditional.py:165-167); 'uniform' = U(-1/sqrt(fan_in), 1/sqrt(fan_in)), the
    distribution family of torch's default Linear init, which gives non-trivial s and t."""
    out = {}
    keys = list(shapes.keys())
    for key in keys:
        shp = tuple(shapes[key])
        if init == "randn":
            out[key] = (0.005 * rs.standard_normal(shp)).astype(np.float32)
        else:
            if len(shp) == 2:
                fan_in = shp[1]
            else:
                fan_in = tuple(shapes[key.replace("bias", "weight")])[1]
            bound = wscale / math.sqrt(max(fan_in, 1))
            out[key] = rs.uniform(-bound, bound, size=shp).astype(np.float32)
    return out


def make_block_inputs(case, shapes):
    c = norm_case(case)
    rs = np.random.RandomState(_seed_of(c["name"]))
    params = draw_params(shapes, c["init"], rs, c["wscale"])
    x = rs.standard_normal((c["B"], c["d"])).astype(np.float32)
    conds = [rs.standard_normal((c["B"], dc[0])).astype(np.float32) for dc in c["dims_c"]]
    return params, x, conds


def make_chain_inputs(case, shapes):
    rs = np.random.RandomState(_seed_of(case["name"]))
    params = [draw_params(shapes, "randn" if case["steps"] == 0 else "uniform", rs)
              for _ in range(case["n_blocks"])]
    perms = []
    for i in range(case["n_blocks"]):
        if i == 0:
            perms.append(None)
        else:
            q, r = np.linalg.qr(rs.standard_normal((case["d"], case["d"])))
            perms.append((q * np.sign(np.diag(r))).astype(np.float32))
    xs = [rs.standard_normal((case["B"], case["d"])).astype(np.float32)
          for _ in range(max(1, case["steps"]))]
    return params, perms, xs


def checksum(arrays) -> float:
    tot = 0.0
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        tot += float(np.sum(a * np.cos(np.arange(a.size).reshape(a.shape) * 0.37)))
    return tot
