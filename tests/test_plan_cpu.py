"""Planner (hint_amd/csrc/hint_plan.cpp) without a GPU: hint_plan_check builds the launch plan of a block
on the host, lets the planner verify its own schedule (every fragment tile of every group in exactly one
wavefront's range of either GEMM phase, slices consistent with the ranges; every scatter target and every
transformed lane of a backward boundary in exactly one slot) and reports the plan's figures.  Covers the BASELINE configs' block shapes, the conditional lanes, wide
(h = 512) nodes and ragged widths."""
import ctypes as C

import pytest
import torch

import hint_amd
from hint_amd import _lib
from hint_amd.hint import node_descs

STAT = ["groups", "levels", "WT", "ST", "lds_fwd", "lds_bwd", "nw", "wjobs", "params", "packed", "units", "abuf_tiles",
        "sub_groups", "wave_local", "small_jobs", "max_slots"]


def check(tree, d, dc, clamp=4.0):
    lib = _lib.load()
    nodes = tree._flat_nodes()
    descs, params, offsets, total = node_descs(nodes)
    stats = (C.c_int64 * 16)()
    st = lib.hint_plan_check(descs, len(nodes), d, dc, clamp, stats)
    assert st == 0, lib.hint_last_error().decode()
    return dict(zip(STAT, list(stats))), nodes, total


@pytest.mark.parametrize("d,dc,widths,n_nodes,n_levels", [
    (6, 0, [200, 100, 50, 25], 3, 2),        # cfg 1 (power_hint_4.py); nodes / levels as SURVEY.md §8a counts them
    (6, 0, [140, 70, 35, 17], 3, 2),         # cfg 2 (power_hint_8.py)
    (8, 0, [128, 64, 32, 16], 7, 3),         # cfg 3 (gas_hint_8.py)
    (100, 0, [224, 112, 56], 71, 7),         # cfg 4 x lane (conditional_hint_4_full.py)
    (100, 4, [224, 112, 56], 71, 7),         # conditional_recursive_cinn_4.py style
    (43, 0, [67, 33, 16, 8], 31, 5),         # cfg 5 as BASELINE words it
    (42, 0, [67, 33, 16, 8], 31, 5),         # the reference's MINIBOONE width (data.py:423)
    (6, 0, [512, 256, 128], 3, 2),           # *_big: h = 512
    (100, 0, [512, 256, 128, 64], None, None),   # plus_hint_4_big (bench.py): the widest root of the workloads
    (5, 0, [385], None, None),
    (1, 0, [8], 1, 1),
    (2, 3, [7, 5], 1, 1),
    (128, 0, [32, 16], None, None),
    (9, 2, [19, 11, 3], 7, 3),               # ragged everything
])
def test_planner_covers_every_tile_exactly_once(d, dc, widths, n_nodes, n_levels):
    dims_c = [(dc,)] if dc else []
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], dims_c=dims_c, c_internal=widths)
    st, nodes, total = check(blk.tree, d, dc)
    if n_nodes is not None:
        assert len(nodes) == n_nodes and st["levels"] == n_levels
    assert st["levels"] == 1 + max(depth for _, _, depth in nodes)
    assert total == sum((p.numel() + 3) // 4 * 4 for p in blk.parameters())
    assert total - 4 < st["params"] <= total          # (the last tensor's padding is not part of the plan)
    assert 0 < st["lds_fwd"] <= 160 * 1024 and 0 < st["lds_bwd"] <= 160 * 1024
    assert st["groups"] >= st["levels"]
    assert st["units"] == 2 * len(nodes) and st["nw"] in (4, 8, 16)
    # activation columns: every (node, net) once, h padded to 16; coupling-gradient columns: r padded to 4
    assert st["WT"] == sum(2 * ((n.s[0].out_features + 15) // 16 * 16) for n, _, _ in nodes)
    assert st["ST"] == sum(2 * ((n.s[4].out_features + 3) // 4 * 4) for n, _, _ in nodes)
    # part B: tiles of up to 48 x 48 outputs of every weight matrix (dW1 in a lane part and a condition part)
    # lean groups (every node of a depth has 1..4 inputs and <= 4 outputs, no condition; outputs staged in LDS):
    # their dW1 / db1 come from the backward kernel, not from part-B jobs
    node_lean = lambda n: dc == 0 and 1 <= n.split_idx <= 4 and n.s[4].out_features <= 4
    lean_depth = {}
    for n, _, depth in nodes:
        lean_depth[depth] = lean_depth.get(depth, True) and node_lean(n)
    t3 = lambda v: ((v + 15) // 16 + 2) // 3
    jobs, jobs_unfused = 0, 0
    for n, _, depth in nodes:
        h, r, k = n.s[0].out_features, n.s[4].out_features, n.split_idx
        base = t3(h) * t3(h) + t3(r) * t3(h) + (t3(h) * t3(dc) if dc else 0) + (t3(h) if (k + dc) == 0 else 0)
        dw1 = t3(h) * t3(k) if k else 0
        jobs += 2 * (base + (0 if lean_depth[depth] else dw1))
        jobs_unfused += 2 * (base + dw1)
    # (the fusion needs the group's outputs staged in LDS: groups too large for that keep their dW1 jobs)
    assert jobs <= st["wjobs"] <= jobs_unfused
    assert st["abuf_tiles"] >= max(2 * ((n.s[0].out_features + 15) // 16) for n, _, _ in nodes)


def test_planner_accepts_the_conditional_lane_couplings():
    """ExternalAffineCoupling (all of x transformed, conditioned on y: k = 0) and the y lane's AffineCoupling"""
    flow = hint_amd.ConditionalHintFlow(10, 3, 1, 24)
    for m, d, dc in ((flow.ac_y_to_x[0], 10, 3), (flow.ac_y[0], 3, 0), (flow.hac_x[0], 10, 0)):
        st, nodes, _ = check(m.tree, d, dc)
        assert st["groups"] >= 1 and st["lds_bwd"] <= 160 * 1024


def test_planner_rejects_malformed_trees():
    lib = _lib.load()
    blk = hint_amd.HierarchicalAffineCouplingBlock([(6,)], c_internal=[16, 8])
    nodes = blk.tree._flat_nodes()
    descs, _, _, _ = node_descs(nodes)
    stats = (C.c_int64 * 16)()
    descs[1].off = 1                       # overlaps its sibling
    assert lib.hint_plan_check(descs, len(nodes), 6, 0, 4.0, stats) != 0
    assert b"overlap" in lib.hint_last_error()
    descs, _, _, _ = node_descs(nodes)
    descs[0].r = 2                         # r != D - k
    assert lib.hint_plan_check(descs, len(nodes), 6, 0, 4.0, stats) != 0
    assert lib.hint_plan_check(descs, len(nodes), 600, 0, 4.0, stats) != 0     # more lanes than the kernels take


@pytest.mark.parametrize("d,widths,sub_groups,wave_local", [
    (43, [67, 33, 16, 8], 2, 0),             # cfg 5: depth 3 and 4 (8 subtrees of 3 nodes, hidden width 8) run one subtree per wavefront
    (42, [67, 33, 16, 8], 2, 0),
    (100, [224, 112, 56], 0, 0),             # cfg 4: its deep levels are 56 wide (four tiles): general groups
    (100, [48, 24, 20, 12, 8, 8], None, 0),  # D = 3 leaves one level above the deepest D = 2 nodes: some wavefronts have an empty level
    (26, [16, 16, 8, 8, 8], 3, 0),
    (6, [140, 70, 35, 17], 0, 1),            # cfg 2: the whole tree on the wave-local kernels instead
    (8, [128, 64, 32, 16], 0, 1),
])
def test_planner_subtree_groups(d, widths, sub_groups, wave_local):
    """which trees get subtree groups (hint_sub.hpp): the planner checks their invariants itself (one row per unit, the
    wavefronts' units in order, the tape lanes a partition of the block's lanes) and fails the plan otherwise"""
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=widths)
    st, nodes, _ = check(blk.tree, d, 0)
    assert st["wave_local"] == wave_local
    if sub_groups is not None:
        assert st["sub_groups"] == sub_groups
    else:
        assert 1 <= st["sub_groups"] <= 3
    # part B's single-tile jobs share workgroups (one per wavefront): all of them in trees with subtree groups, those that rebuild
    # their operands in every tree on the general kernels, none in the narrow trees of the wave-local kernels
    if st["sub_groups"]:
        assert st["small_jobs"] > 0 and st["groups"] > st["sub_groups"]
    elif wave_local:
        assert st["small_jobs"] == 0


@pytest.mark.parametrize("d,widths,slots", [
    (6, [140, 70, 35, 17], 4),        # cfg 2: the boundary between the levels has 5 active lanes - 2 both, 2 coupling-only, 1 scatter-only
    (8, [128, 64, 32, 16], 4),        # cfg 3: 6 active lanes, 2 + 2 + 2
    (43, [67, 33, 16, 8], 27),        # cfg 5: up to 33-38 active lanes of 43 (16 x 33 were two passes of the backward's 512 threads)
])
def test_backward_boundary_slots_pair_coupling_only_with_scatter_only_lanes(d, widths, slots):
    """hint_plan.cpp (slot table): a thread of a backward boundary takes a lane that is scatter target AND transformed, or one of each
    kind, so the widest boundary has fewer slots than the tree has lanes - and one pass of the workgroup covers it (the planner's
    self-check, run by hint_plan_check, proves that every lane's work is in exactly one slot)."""
    blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], dims_c=[], c_internal=widths)
    st, nodes, _ = check(blk.tree, d, 0)
    assert st["max_slots"] == slots < d
    assert 16 * st["max_slots"] <= 64 * st["nw"] or st["wave_local"]


@pytest.mark.parametrize("env", [{"HINT_LEANW": "0"}, {}, {"HINT_LEANW_MAX": "28"}])
def test_lean_wide_knobs_change_no_layout(monkeypatch, env):
    """lean-wide groups (Group::lean bit 3: a1 / g2 of thin layers with 5 .. HINT_LEANW_MAX inputs / outputs rebuilt by part B) only
    switch stores off and re-source the dW2 jobs: arrays, parameters, jobs and LDS of a plan are the same with and without them,
    and the knobs that are set show up in hint_build_info()"""
    lib = _lib.load()
    for k in ("HINT_LEANW", "HINT_LEANW_MAX"):
        monkeypatch.delenv(k, raising=False)
    lib.hint_debug_reload_knobs()
    base = {}
    for d, widths in ((43, [67, 33, 16, 8]), (100, [224, 112, 56])):
        blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=widths)
        base[d] = check(blk.tree, d, 0)[0]
    try:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lib.hint_debug_reload_knobs()
        info = lib.hint_build_info().decode()
        for k, v in env.items():
            assert f"{k}={v}" in info
        assert ("knobs:" in info) == bool(env)
        for d, widths in ((43, [67, 33, 16, 8]), (100, [224, 112, 56])):
            blk = hint_amd.HierarchicalAffineCouplingBlock([(d,)], c_internal=widths)
            got = check(blk.tree, d, 0)[0]
            # (single-tile dW2 jobs that rebuild their operands share workgroups, one per wavefront: their count follows the knob)
            assert {k: v for k, v in got.items() if k != "small_jobs"} == {k: v for k, v in base[d].items() if k != "small_jobs"}
    finally:
        for k in env:
            monkeypatch.delenv(k, raising=False)
        lib.hint_debug_reload_knobs()
