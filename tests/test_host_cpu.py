"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the module mirror has the reference's structure and state_dict keys, and the product
path refuses to run without a GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

import hint_amd
from hint_amd import _lib
from oracle import hint_oracle as orc
from util import BLOCK_CASES, load_block_case, norm_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "hint_amd.h")).read()
    declared = set(re.findall(r"\b(hint_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hint_abi_version() == 7


def test_nodedesc_matches_header_layout():
    # 6 x int32 + 12 x int64 = 120 bytes, no padding surprises
    import ctypes as C
    assert C.sizeof(_lib.NodeDesc) == 6 * 4 + 12 * 8


@pytest.mark.parametrize("case", BLOCK_CASES, ids=lambda c: c["name"])
def test_state_dict_contract(case):
    """same keys, order and shapes as the reference module (pinned in the golden fixtures)"""
    c, nodes, shapes, params, x, conds, g = load_block_case(case)
    blk = hint_amd.HierarchicalAffineCouplingBlock([(c["d"],)], dims_c=c["dims_c"], c_internal=list(c["c_internal"]),
                                                   clamp=c["clamp"], max_splits=c["max_splits"],
                                                   min_split_size=c["min_split_size"], reshuffle=c["reshuffle"])
    sd = blk.state_dict()
    perm_keys = [k for k in sd if k.endswith(".perm.W")]       # stand-in for FrEIA's HouseholderPerm state
    assert bool(perm_keys) == c["reshuffle"]
    assert [k for k in sd if k not in perm_keys] == list(g["keys"])
    for k, v in sd.items():
        if k in perm_keys:
            W = v.double()
            assert torch.allclose(W @ W.t(), torch.eye(W.shape[0], dtype=torch.float64), atol=1e-5)
        else:
            assert tuple(v.shape) == tuple(shapes[k])
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=not c["reshuffle"])
    # flat node list agrees with the oracle's restatement of hint.py:25-54
    flat = blk.tree._flat_nodes()
    assert [(o, n.data_shape[0], n.split_idx, d, n.leaf) for n, o, d in flat] == \
           [(n.off, n.D, n.k, n.depth, n.leaf) for n in nodes]


def test_c_internal_not_mutated_and_defaults():
    w = [12]
    blk = hint_amd.HierarchicalAffineCouplingBlock([(9,)], c_internal=w)
    assert w == [12]          # the reference doubles the caller's list in place (hint.py:33-34)
    assert blk.tree.s[0].out_features == 12 and blk.tree.upper.s[0].out_features == 12
    assert blk.output_dims([(9,)]) == [(9,)]
    with pytest.raises(AssertionError):
        blk.output_dims([(9,), (9,)])


def test_unsupported_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        hint_amd.HierarchicalAffineCouplingBlock([(6,)], conv=True)
    with pytest.raises(NotImplementedError):
        hint_amd.HierarchicalAffineCouplingBlock([(6,)], subnet_constructor=lambda a, b, c: None)
    with pytest.raises(NotImplementedError):          # a subnet of another shape than hint.py:10-13
        hint_amd.HierarchicalAffineCouplingBlock([(6,)], subnet_constructor=lambda a, b, c: torch.nn.Sequential(torch.nn.Linear(a, b)))


def test_custom_subnet_constructor_of_the_reference_shape_is_taken():
    """hint.py:27-32: subnet_constructor(c_in, c_out, c_internal) is called per subnet; one that returns the reference's own
    Linear-ReLU-Linear-ReLU-Linear (here with another initialisation) is adopted - same state_dict keys, its own weights"""
    calls = []

    def make(c_in, c_out, c_internal):
        calls.append((c_in, c_out, c_internal))
        net = hint_amd.linear_subnet_constructor(c_in, c_out, c_internal)
        for p in net.parameters():
            torch.nn.init.constant_(p, 0.25)
        return net

    blk = hint_amd.HierarchicalAffineCouplingBlock([(6,)], subnet_constructor=make, c_internal=[16, 8])
    ref = hint_amd.HierarchicalAffineCouplingBlock([(6,)], c_internal=[16, 8])
    assert list(blk.state_dict().keys()) == list(ref.state_dict().keys())
    assert calls[0] == (3, 3, 16) and calls[1] == (3, 3, 16) and (1, 2, 8) in calls and len(calls) == 6
    assert all(float(v.min()) == 0.25 == float(v.max()) for v in blk.state_dict().values())


def test_reshuffle_composes_to_one_orthogonal_matrix():
    """reshuffle=True (hint.py:36-39): every node owns a fixed orthogonal matrix; the engine folds
    them (top-down, block-diagonal per level) into the one [d,d] matrix the kernels apply"""
    torch.manual_seed(3)
    blk = hint_amd.HierarchicalAffineCouplingBlock([(9,)], c_internal=[8, 4, 4], reshuffle=True)
    mods = [(n, o, d) for n, o, d in blk.tree._flat_nodes()]
    assert all(n.perm is not None for n, _, _ in mods)
    x = torch.randn(5, 9, dtype=torch.float64)
    y = x.clone()
    for n, off, depth in sorted(mods, key=lambda t: t[2]):          # what the recursion does, level by level
        D = n.data_shape[0]
        y[:, off:off + D] = y[:, off:off + D] @ n.perm.W.double()
    tot = torch.eye(9, dtype=torch.float64)
    for n, off, depth in sorted(mods, key=lambda t: t[2]):
        D = n.data_shape[0]
        b = torch.eye(9, dtype=torch.float64)
        b[off:off + D, off:off + D] = n.perm.W.double()
        tot = tot @ b
    assert torch.allclose(x @ tot, y, atol=1e-12)
    assert torch.allclose(tot @ tot.t(), torch.eye(9, dtype=torch.float64), atol=1e-5)


def test_no_cpu_fallback():
    blk = hint_amd.HierarchicalAffineCouplingBlock([(6,)], c_internal=[8, 4])
    with pytest.raises(hint_amd.HintAmdError):
        blk([torch.randn(4, 6)])


def test_product_code_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hint_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU oracle", ""), os.path.join(dirpath, f)


def test_conditional_lane_modules_structure():
    """conditional_hint_4_full.py:55-95: the two-lane model and its couplings are single-node trees
    of the same kernels (AffineCoupling: split at D // 2; ExternalAffineCoupling: empty upper half)"""
    m = hint_amd.ConditionalHintFlow(100, 4, 4, 224)
    assert len(m.hac_x) == len(m.ac_y_to_x) == len(m.ac_y) == 4
    e = m.ac_y_to_x[0].tree
    assert e.leaf and e.split_idx == 0 and e.s[0].in_features == 4 and e.s[4].out_features == 100
    a = m.ac_y[1].tree
    assert a.leaf and a.split_idx == 2 and a.s[0].in_features == 2 and a.s[0].out_features == 112
    assert [w for w in (m.hac_x[0].tree.s[0].out_features, m.hac_x[0].tree.upper.s[0].out_features)] == [224, 112]
    assert isinstance(m.perm_x[0], torch.nn.Identity) and hasattr(m.perm_x[1], "W")
    with pytest.raises(hint_amd.HintAmdError):
        hint_amd.ExternalAffineCoupling([(5,)])                  # needs a condition
    with pytest.raises(hint_amd.HintAmdError):
        m([torch.randn(3, 4), torch.randn(3, 100)])              # CPU tensors: no fallback


def test_data_alias_edits_do_not_advance_the_version_counter():
    """why hint_amd.hint.set_pack_cache is opt-in: an in-place edit through `p.data` - `p.data.add_(...)`, `p.data.clamp_(...)` -
    changes the weights without touching `p._version` or `p.data_ptr()`, so no host-side key can prove a packed copy current;
    optimizer updates (in place under no_grad) and `p.data = ...` rebinding are visible"""
    import torch
    p = torch.nn.Parameter(torch.randn(8))
    v0, a0 = p._version, p.data_ptr()
    p.data.add_(1.0)
    assert (p._version, p.data_ptr()) == (v0, a0)
    with torch.no_grad():
        p.add_(1.0)
    assert p._version == v0 + 1
    p.data = torch.randn(8)
    assert p.data_ptr() != a0
