"""Host-side mirror of the reference's `hint.py` module surface, backed by the HIP library.

Drop-in for /root/reference/hint.py:
  linear_subnet_constructor ............ hint.py:10-13
  HierarchicalAffineCouplingTree ....... hint.py:21-101  (same ctor keywords, same attributes
                                         split_idx / conditional / leaf / s / t / upper / lower,
                                         hence the same state_dict keys `tree.s.0.weight` ...)
  HierarchicalAffineCouplingBlock ...... hint.py:104-133 (FrEIA module protocol: list in / list
                                         out, cached `self.jac`, jacobian(), output_dims())
The arithmetic itself runs in hint_amd/csrc (gfx950 kernels) through the C ABI of
include/hint_amd.h.  There is no CPU implementation here: CPU tensors raise.

Parameters stay ordinary `nn.Parameter`s inside `nn.Linear`s (the training loop rebinds
`p.data`, clamps `p.grad`, hands them to Adam: train_unconditional.py:140-141,165-176).  For
the kernels they are kept as views into ONE flat fp32 arena per tree; the arena is
re-gathered automatically whenever a parameter stops aliasing it (`p.data = ...`, `.to()`).
"""
from __future__ import annotations

import ctypes as C
import operator
import os
import time
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from ._lib import HintAmdError, NodeDesc

__all__ = ["linear_subnet_constructor", "HierarchicalAffineCouplingTree",
           "HierarchicalAffineCouplingBlock", "HintAmdError", "set_param_grad_mode", "set_pack_cache"]

_ALIGN = 4  # floats; every tensor starts 16-byte aligned inside the arena

# ---- how parameter gradients reach `p.grad` on the module (autograd) route ------------------------------------------
# "direct" (default): the backward kernels write the flat gradient of a block (of the whole flow on HintFlow's fused
#   route) into ONE persistent buffer and every `p.grad` is a cached view into it - what `loss.backward()` leaves for the
#   reference loop's `p.grad.data.clamp_` / `optim.step()` (train_unconditional.py:137-144), without 36 autograd edges,
#   views and AccumulateGrad nodes per block and pass (they cost 10x the kernels at cfg 2).  The usual accumulation rules
#   hold (`p.grad is None` -> set; a gradient left by an earlier backward -> added to).  What does NOT happen: hooks on the
#   parameters (`p.register_hook`, DistributedDataParallel's reducer) do not fire and `torch.autograd.grad(loss, params)`
#   does not see the parameters.
# "autograd": every parameter is an input of the autograd node and its gradient is returned through autograd (the
#   behaviour of rounds 1-5) - for code that needs the hooks or the functional API.
_PARAM_GRADS = os.environ.get("HINT_PARAM_GRADS", "direct")
# Re-pack skipping is OFF by default: `p.data.add_(...)`-style edits do not advance `p._version` (a `.data` alias has a
# version counter of its own), so no host-side key can prove the packed copy current; the re-pack is one 6 us launch.
# set_pack_cache(True) / HINT_PACK_CACHE=1 skips it when no parameter's (`data_ptr`, `_version`) moved since the last pack -
# for loops that only ever change weights through optimizers / load_state_dict / `p.data = ...`.
_PACK_CACHE = os.environ.get("HINT_PACK_CACHE", "0") not in ("", "0")


def set_param_grad_mode(mode: str) -> str:
    """'direct' or 'autograd' (see above); returns the previous mode.  Takes effect at the next forward."""
    global _PARAM_GRADS
    if mode not in ("direct", "autograd"):
        raise ValueError("mode must be 'direct' or 'autograd'")
    prev, _PARAM_GRADS = _PARAM_GRADS, mode
    return prev


def set_pack_cache(on: bool) -> bool:
    global _PACK_CACHE
    prev, _PACK_CACHE = _PACK_CACHE, bool(on)
    return prev


# ---- where a step's time goes (bench.py's module_path entry): host clock + HIP events around hint_amd's entry points ----
_PROF = None


def profile_start():
    global _PROF
    _PROF = {"host": 0.0, "events": [], "calls": 0, "cur": None}
    return _PROF


def profile_stop(prof, n_steps: int):
    global _PROF
    _PROF = None
    torch.cuda.synchronize()
    dev_ms = sum(a.elapsed_time(b) for a, b in prof["events"])
    return {"hint_amd_host_ms": prof["host"] * 1e3 / n_steps, "hint_amd_device_ms": dev_ms / n_steps,
            "hint_amd_calls_per_step": prof["calls"] / n_steps}


class _Region:
    """`with _Region():` around an entry point: nothing unless profile_start() is active.  Host clock over the whole entry; the
    HIP events bracket its LAUNCHES (first one: _mark_launch() at the launch sites - the host-side preparation in front of it is
    host time, not device time)"""
    __slots__ = ("pr", "t0", "e0")

    def __enter__(self):
        self.pr = _PROF
        if self.pr is not None:
            self.e0 = None
            self.pr["cur"] = self
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        pr = self.pr
        if pr is not None:
            pr["host"] += time.perf_counter() - self.t0
            if self.e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                pr["events"].append((self.e0, e1))
            pr["cur"] = None
            pr["calls"] += 1
        return False


def _mark_launch():
    """the first launch of the current entry point is about to go out (profile_start() only)"""
    pr = _PROF
    if pr is not None:
        cur = pr.get("cur")
        if cur is not None and cur.e0 is None:
            cur.e0 = torch.cuda.Event(enable_timing=True)
            cur.e0.record()


class _Lease:
    """hands a pooled buffer back when the autograd node that holds it dies (not at the end of backward():
    retain_graph=True may run the node again); a pool keeps two - what a third one owns besides memory is released by `drop`"""
    __slots__ = ("pool", "item", "drop")

    def __init__(self, pool, item, drop=None):
        self.pool, self.item, self.drop = pool, item, drop

    def __del__(self):
        try:
            if len(self.pool) < 2:
                self.pool.append(self.item)
            elif self.drop is not None:
                self.drop(self.item)
        except Exception:
            pass


def linear_subnet_constructor(c_in, c_out, c_internal):
    """Linear-ReLU-Linear-ReLU-Linear, the only subnet the kernels implement (hint.py:10-13)."""
    return nn.Sequential(nn.Linear(c_in, c_internal), nn.ReLU(),
                         nn.Linear(c_internal, c_internal), nn.ReLU(),
                         nn.Linear(c_internal, c_out))


def _checked_subnet(net, c_in, c_out):
    """a subnet the HIP kernels can run: nn.Sequential(Linear(c_in, h), ReLU, Linear(h, h), ReLU, Linear(h, c_out)), all with
    bias (what linear_subnet_constructor builds, hint.py:10-13); NotImplementedError otherwise"""
    # (exactly nn.Linear / nn.ReLU: a subclass with a forward of its own - LoRA, quantised - would be computed as a plain layer)
    ok = isinstance(net, nn.Sequential) and len(net) == 5 and all(type(net[i]) is nn.Linear for i in (0, 2, 4)) \
        and all(type(net[i]) is nn.ReLU for i in (1, 3)) and all(net[i].bias is not None for i in (0, 2, 4)) \
        and all(net[i].weight.dtype == torch.float32 for i in (0, 2, 4))
    if ok:
        h = net[0].out_features
        ok = net[0].in_features == c_in and net[2].in_features == h and net[2].out_features == h \
            and net[4].in_features == h and net[4].out_features == c_out
    if not ok:
        raise NotImplementedError("subnet_constructor must return nn.Sequential(Linear(c_in, h), ReLU(), Linear(h, h), ReLU(), "
                                  "Linear(h, c_out)) - the subnet of hint.py:10-13 is the only one implemented in HIP")
    return net


def conv_subnet_constructor(c_in, c_out, c_internal):
    raise NotImplementedError("conv subnets (hint.py:15-18) are out of scope: no reference config uses conv=True")


def node_descs(nodes):
    """the C ABI's node table (hint_node_desc[]) of a flattened tree, its parameters in arena order,
    their float offsets and the arena size: what hint_plan_create / hint_plan_check take"""
    params: List[nn.Parameter] = []
    offsets: List[int] = []
    descs = (NodeDesc * len(nodes))()
    cursor = 0
    for i, (node, off, depth) in enumerate(nodes):
        D = node.data_shape[0]
        dsc = descs[i]
        dsc.off, dsc.D, dsc.k, dsc.r = off, D, node.split_idx, D - node.split_idx
        dsc.h, dsc.depth = node.s[0].out_features, depth
        j = 0
        for net in (node.s, node.t):
            for li in (0, 2, 4):
                for p in (net[li].weight, net[li].bias):
                    dsc.p_off[j] = cursor
                    params.append(p)
                    offsets.append(cursor)
                    cursor += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                    j += 1
    return descs, params, offsets, cursor


class _Engine:
    """Plan + flat parameter arena of one (root) tree on one device."""

    def __init__(self, tree: "HierarchicalAffineCouplingTree", device: torch.device):
        self.lib = _lib.load()
        self.device = device
        self.d = tree.data_shape[0]
        self.dc = tree.condition_length
        nodes = tree._flat_nodes()
        descs, self.params, self.offsets, cursor = node_descs(nodes)
        self.total = max(cursor, _ALIGN)
        # two trees with equal keys get identical plans (they can share chained launches)
        self.shape_key = (self.d, self.dc, float(tree.clamp),
                          tuple((n.off, n.D, n.k, n.r, n.h, n.depth, tuple(n.p_off)) for n in descs))
        self.shapes = [tuple(p.shape) for p in self.params]
        self.numels = [p.numel() for p in self.params]
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.hint_plan_create(descs, len(nodes), self.d, self.dc, float(tree.clamp),
                                                 C.byref(handle)), "hint_plan_create")
        self.plan = handle
        self.arena: Optional[torch.Tensor] = None
        self.packed: Optional[torch.Tensor] = None
        self._ext_arena: Optional[torch.Tensor] = None
        self._ptrs: List[int] = []
        self._regathered = False
        self._sizes = {}            # B -> (tape floats, workspace bytes) of this plan
        self._tape_pool = {}        # (B, stream) -> free tape buffers (handed back when the autograd node that used one dies)
        self._ws = {}               # (B, stream) -> the backward workspace (scratch inside one stream-ordered call)
        self._pack_key = None       # (data_ptr, _version) of every parameter at the last pack (set_pack_cache)
        self._gathers = 0           # times the arena was re-gathered (a gather can restore every data_ptr with new contents)
        self._gflat: Optional[torch.Tensor] = None      # 'direct' parameter gradients: the block's flat gradient ...
        self._gviews: List[torch.Tensor] = []           # ... and one cached view per parameter (what p.grad becomes)
        self._anchor: Optional[torch.Tensor] = None     # the autograd edge that makes backward() run in 'direct' mode
        # reshuffle=True: the per-node matrices act top-down before any coupling of their subtree
        # (hint.py:64-65), so all of them compose into ONE [d,d] orthogonal matrix in front of the
        # block (and its transpose behind the inverse), which the kernels apply fused
        self._perm_nodes = [(node.perm, off, node.data_shape[0], depth) for node, off, depth in nodes
                            if node.perm is not None]
        self._perm_total: Optional[torch.Tensor] = None
        self._perm_key = None

    def total_perm(self) -> Optional[torch.Tensor]:
        """the block's composed node permutations as a contiguous [d,d] device tensor (None if the tree
        has none); recomputed when a node's matrix was replaced (load_state_dict, .to())"""
        if not self._perm_nodes:
            return None
        key = tuple((m.W.data_ptr(), m.W._version) for m, _, _, _ in self._perm_nodes)
        if key != self._perm_key or self._perm_total is None:
            tot = torch.eye(self.d, dtype=torch.float64)
            for m, off, D, _ in sorted(self._perm_nodes, key=lambda t: t[3]):
                blk = torch.eye(self.d, dtype=torch.float64)
                blk[off:off + D, off:off + D] = m.W.detach().double().cpu()
                tot = tot @ blk
            self._perm_total = tot.to(torch.float32).contiguous().to(self.device)
            self._perm_key = key
        return self._perm_total

    def compose_perm(self, front: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """permutation in front of the block for the chained entry points: `front` (a flow's fixed
        inter-block matrix, or None) followed by the block's own node permutations"""
        own = self.total_perm()
        if own is None:
            return front
        if front is None:
            return own
        key = (front.data_ptr(), front._version, own.data_ptr())
        if getattr(self, "_composed_key", None) != key:       # cached: callers key on its address
            self._composed = (front.double() @ own.double()).to(torch.float32).contiguous()
            self._composed_key = key
        return self._composed

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.hint_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # ---- arena management -------------------------------------------------------------
    def bind_external_arena(self, arena: torch.Tensor):
        """Use a caller-owned flat buffer (e.g. a slice of a model-wide arena)."""
        assert arena.dtype == torch.float32 and arena.is_contiguous() and arena.numel() >= self.total
        self._ext_arena = arena
        self.arena = None
        self._ptrs = []

    def _gather(self):
        arena = self._ext_arena
        if arena is None or arena.device != self.device:
            arena = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for p, off, n, shp in zip(self.params, self.offsets, self.numels, self.shapes):
                if p.dtype != torch.float32:
                    raise HintAmdError("hint_amd kernels are fp32 only (parameter dtype %s)" % p.dtype)
                view = arena[off:off + n].view(shp)
                if p.data.data_ptr() != view.data_ptr() or p.device != self.device:
                    view.copy_(p.data.to(self.device))
                    p.data = view
        self.arena = arena
        self._regathered = True
        self._gathers += 1
        base = arena.data_ptr()
        self._ptrs = [base + 4 * off for off in self.offsets]

    def ensure_arena(self):
        # (one list comparison: 36 data_ptr() calls are 2.4 us; `p.data = ...` / `.to()` cannot be seen any cheaper)
        if self.arena is None or [p.data_ptr() for p in self.params] != self._ptrs:
            self._gather()

    def sizes(self, B: int):
        """(tape floats, workspace bytes) for a batch of B rows, asked of the library once per batch size"""
        v = self._sizes.get(B)
        if v is None:
            v = (max(self.lib.hint_plan_tape_floats(self.plan, B), 1),
                 max(int(self.lib.hint_plan_workspace_bytes(self.plan, B)), 16))
            if len(self._sizes) > 64:
                self._sizes.clear()
            self._sizes[B] = v
        return v

    def take_tape(self, B: int, device):
        """-> (tape, lease): a tape buffer for one training forward.  Buffers are pooled per (batch size, stream); the lease
        object belongs on the autograd node - when the node dies the buffer returns to the pool (a forward that runs while an
        older node of this block is still alive simply gets another buffer)"""
        n = self.sizes(B)[0]
        key = (B, torch.cuda.current_stream(device).cuda_stream)
        pool = self._tape_pool.get(key)
        if pool is None:
            if len(self._tape_pool) >= 4:          # ragged batch sizes: keep the pools of the latest ones only
                self._tape_pool.clear()
            pool = self._tape_pool[key] = []
        tape = pool.pop() if pool else torch.empty(n, dtype=torch.float32, device=device)
        return tape, _Lease(pool, tape)

    def workspace(self, B: int, device):
        nbytes = self.sizes(B)[1]
        key = (B, torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4:
                self._ws.clear()
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return ws, nbytes

    # ---- 'direct' parameter gradients ---------------------------------------------------------
    def anchor(self) -> torch.Tensor:
        if self._anchor is None or self._anchor.device != self.device:
            self._anchor = torch.zeros(1, dtype=torch.float32, device=self.device, requires_grad=True)
        return self._anchor

    def grad_views(self):
        if self._gflat is None or self._gflat.device != self.device:
            self._gflat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self._gviews = self.split_flat(self._gflat)
        return self._gflat, self._gviews

    def invalidate_packed(self):
        """the packed copy no longer matches the arena (a trainer's kernels updated the weights in place)"""
        self._pack_key = None

    def split_flat(self, flat: torch.Tensor):
        return [flat[off:off + n].view(shp) for off, n, shp in zip(self.offsets, self.numels, self.shapes)]

    # ---- kernel launches --------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def pack(self):
        """(re)build the MFMA-fragment-order copy of the weights the kernels read.  Cheap (one
        small launch), so the module path calls it on every forward; FlowTrainer calls it once
        per optimizer step."""
        if self.packed is None or self.packed.device != self.device:
            self.packed = torch.empty(self.lib.hint_plan_packed_floats(self.plan), dtype=torch.float32,
                                      device=self.device)
            self._pack_key = None
        _mark_launch()
        if _PACK_CACHE:
            key = [self._gathers] + [(p.data_ptr(), p._version) for p in self.params]
            if key == self._pack_key:
                return
            self._pack_key = key
        with torch.cuda.device(self.device):
            st = self.lib.hint_block_pack(self.plan, self.arena.data_ptr(), self.packed.data_ptr(), self._stream())
        _lib.check(st, "hint_block_pack")

    def apply(self, x: torch.Tensor, c: Optional[torch.Tensor], rev: bool, with_tape: bool = False, lease=None):
        """-> (out, J) or, with_tape (training forward), (out, J, tape); lease: a list that receives the pooled tape's
        lease object (to be kept on the autograd node) - without it the tape is a plain allocation"""
        B = x.shape[0]
        out = torch.empty_like(x)
        J = torch.empty(B, dtype=torch.float32, device=x.device)
        tape = None
        if with_tape:
            if lease is not None:
                tape, ls = self.take_tape(B, x.device)
                lease.append(ls)
            else:
                tape = torch.empty(self.sizes(B)[0], dtype=torch.float32, device=x.device)
        if B > 0:
            cptr = c.data_ptr() if c is not None else None
            perm = self.total_perm()
            pptr = perm.data_ptr() if perm is not None else None
            with torch.cuda.device(self.device):
                if rev:
                    st = self.lib.hint_block_inverse_ex(self.plan, self.arena.data_ptr(), self.packed.data_ptr(),
                                                        x.data_ptr(), cptr, out.data_ptr(), J.data_ptr(), pptr, None,
                                                        B, self._stream())
                else:
                    st = self.lib.hint_block_forward_ex(self.plan, self.arena.data_ptr(), self.packed.data_ptr(),
                                                        x.data_ptr(), cptr, out.data_ptr(), J.data_ptr(),
                                                        tape.data_ptr() if tape is not None else None, pptr, None,
                                                        None, B, self._stream())
            _lib.check(st, "hint_block_inverse_ex" if rev else "hint_block_forward_ex")
        return (out, J, tape) if with_tape else (out, J)

    # ---- chained forms (flow container / trainer): permutation, J accumulation, loss fused ----
    def forward_chain(self, x, c, perm, J_in, loss_acc, with_tape: bool):
        B = x.shape[0]
        out = torch.empty_like(x)
        J = torch.empty(B, dtype=torch.float32, device=x.device)
        tape = None
        if with_tape:
            tape = torch.empty(self.sizes(B)[0], dtype=torch.float32, device=x.device)
        if B > 0:
            ptr = lambda t: t.data_ptr() if t is not None else None
            with torch.cuda.device(self.device):
                st = self.lib.hint_block_forward_ex(self.plan, self.arena.data_ptr(), self.packed.data_ptr(),
                                                    x.data_ptr(), ptr(c), out.data_ptr(), J.data_ptr(), ptr(tape),
                                                    ptr(perm), ptr(J_in), ptr(loss_acc), B, self._stream())
            _lib.check(st, "hint_block_forward_ex")
        return out, J, tape

    def backward_chain(self, x, tape, c, gz, gz_scale, gJ_const, perm, g_params, accumulate=True):
        B = gz.shape[0]
        gx = torch.empty_like(gz)
        if B == 0:
            return gx
        ws, nbytes = self.workspace(B, gz.device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):
            st = self.lib.hint_block_backward_ex(
                self.plan, self.arena.data_ptr(), self.packed.data_ptr(), ptr(x), ptr(tape), ptr(c), gz.data_ptr(),
                None, gx.data_ptr(), None, g_params.data_ptr(), 1 if accumulate else 0, ws.data_ptr(), nbytes,
                ptr(perm), float(gz_scale), float(gJ_const), B, self._stream())
        _lib.check(st, "hint_block_backward_ex")
        return gx

    def backward(self, x, tape, c, gz, gJ, need_gc: bool, g_params: Optional[torch.Tensor] = None,
                 accumulate: bool = False, front: Optional[torch.Tensor] = None, gz_scale: float = 1.0,
                 gJ_const: float = 0.0):
        """front: the fixed matrix in front of the block the forward call was given (forward_chain's perm before
        compose_perm), so that g_x comes back through it; gz_scale / gJ_const as in hint_block_backward_ex"""
        B = x.shape[0]
        gx = torch.empty_like(x)
        gc = torch.empty_like(c) if (c is not None and need_gc) else None
        if g_params is None:
            g_params = torch.empty(self.total, dtype=torch.float32, device=x.device)
            accumulate = False
        if B == 0:
            if not accumulate:
                g_params.zero_()
            return gx, gc, g_params
        ws, nbytes = self.workspace(B, x.device)
        perm = self.compose_perm(front)
        _mark_launch()
        with torch.cuda.device(self.device):
            st = self.lib.hint_block_backward_ex(
                self.plan, self.arena.data_ptr(), self.packed.data_ptr(), x.data_ptr(),
                tape.data_ptr() if tape is not None else None, c.data_ptr() if c is not None else None,
                gz.data_ptr() if gz is not None else None, gJ.data_ptr() if gJ is not None else None,
                gx.data_ptr(), gc.data_ptr() if gc is not None else None, g_params.data_ptr(),
                1 if accumulate else 0, ws.data_ptr(), nbytes, perm.data_ptr() if perm is not None else None,
                float(gz_scale), float(gJ_const), B, self._stream())
        _lib.check(st, "hint_block_backward")
        return gx, gc, g_params

    # ---- backward of the INVERSE direction (hint.py:82-88 under autograd) ------------------------------
    def inverse_backward(self, x, c, gx, gJ, need_gc: bool):
        """gradients of (x, J) = block(z, rev=True) given its output x and the upstream g_x [B,d] / g_J [B] (either may be
        None): -> (g_z, g_c or None, flat parameter gradient).  hint_block_inverse_backward runs the tree level by level
        on the block kernels (include/hint_amd.h)."""
        B = x.shape[0]
        gz = torch.empty_like(x)
        gc = torch.empty_like(c) if (c is not None and need_gc) else None
        g_params = torch.empty(self.total, dtype=torch.float32, device=x.device)
        perm = self.total_perm()
        ptr = lambda t: t.data_ptr() if t is not None else None
        _mark_launch()
        with torch.cuda.device(self.device):
            nbytes = self.lib.hint_plan_inverse_workspace_bytes(self.plan, B) if B > 0 else 0
            if B > 0 and nbytes == 0:
                _lib.check(1, "hint_plan_inverse_workspace_bytes")
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
            st = self.lib.hint_block_inverse_backward(self.plan, self.arena.data_ptr(), x.data_ptr(), ptr(c), ptr(gx), ptr(gJ),
                                                      gz.data_ptr(), ptr(gc), g_params.data_ptr(), 0, ws.data_ptr(), nbytes,
                                                      ptr(perm), B, self._stream())
        _lib.check(st, "hint_block_inverse_backward")
        return gz, gc, g_params


import itertools
_NONES = itertools.repeat(None)


def deliver_param_grads(params, views, flat, run, fresh):
    """'direct' mode: make `p.grad` of every parameter hold the gradient `run(g_params, accumulate)` computes (the launch that
    writes the flat gradient of `params` - a block's or a whole flow's - into a buffer of `flat`'s layout).
      every p.grad is None (after optim.zero_grad())        -> overwrite `flat`, p.grad = the cached views
      every p.grad is still our view (no zero_grad between) -> the kernels add to `flat`
      anything else (foreign gradients, a mix)              -> into a fresh buffer, then per parameter as autograd would
    `fresh()` allocates such a buffer and returns (buffer, views)."""
    grads = [p.grad for p in params]
    if all(map(operator.is_, grads, _NONES)):
        run(flat, 0)
        for p, v in zip(params, views):
            p.grad = v
        return
    if all(map(operator.is_, grads, views)):
        run(flat, 1)
        return
    buf, vs = fresh()
    run(buf, 0)
    for p, g, v in zip(params, grads, vs):
        if g is None:
            p.grad = v
        else:
            g.add_(v)


class _RevCouplingFn(torch.autograd.Function):
    """autograd node of one block INVERSE (rev=True with gradients: differentiable in the reference, hint.py:82-88;
    its own loops only sample under no_grad).  Saves the output; the backward rebuilds the tapes level by level
    (_Engine.inverse_backward).  direct: see _PARAM_GRADS (then `params` is the engine's anchor alone)."""

    @staticmethod
    def forward(ctx, engine, direct, z, c, *params):
        x, J = engine.apply(z, c, rev=True)
        ctx.engine = engine
        ctx.direct = direct
        ctx.n_params = len(params)
        ctx.has_c = c is not None
        ctx.save_for_backward(x, c) if c is not None else ctx.save_for_backward(x)
        return x, J

    @staticmethod
    def backward(ctx, gx, gJ):
        with _Region():
            engine = ctx.engine
            saved = ctx.saved_tensors
            x = saved[0]
            c = saved[1] if ctx.has_c else None
            gx = gx.contiguous() if gx is not None else None
            gJ = gJ.contiguous() if gJ is not None else None
            need_gc = ctx.has_c and ctx.needs_input_grad[3]
            engine.ensure_arena()
            gz, gc, gflat = engine.inverse_backward(x, c, gx, gJ, need_gc)
            if not ctx.direct:
                return (None, None, gz, gc, *(engine.split_flat(gflat) if ctx.n_params else ()))
            for p, v in zip(engine.params, engine.split_flat(gflat)):      # (a convenience path: no cached views)
                if p.requires_grad:
                    if p.grad is None:
                        p.grad = v
                    else:
                        p.grad.add_(v)
            return None, None, gz, gc, None


class _CouplingFn(torch.autograd.Function):
    """autograd node of one block forward.  Keeps the input and the tape the forward kernel wrote: the lanes at
    every level, the coupling arguments s, the hidden activations a2 of every subnet and one sign byte per four
    activations (a1 as well, except for the subnets with at most four inputs and outputs: theirs is rebuilt from the
    level's lanes where it is needed - DESIGN.md section 3).  The backward kernels never re-run the h x h layers.
    direct: the parameter gradients go straight to `p.grad` (see _PARAM_GRADS; `params` is the engine's anchor alone)."""

    @staticmethod
    def forward(ctx, engine, direct, x, c, *params):
        lease = []
        z, J, tape = engine.apply(x, c, rev=False, with_tape=True, lease=lease)
        ctx.engine = engine
        ctx.direct = direct
        ctx.n_params = len(params)
        ctx.has_c = c is not None
        ctx.tape, ctx.lease = tape, lease          # (the tape is the library's scratch, not an autograd-visible tensor)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, c) if c is not None else ctx.save_for_backward(x)
        return z, J

    @staticmethod
    def backward(ctx, gz, gJ):
        with _Region():
            engine = ctx.engine
            saved = ctx.saved_tensors
            x, tape = saved[0], ctx.tape
            c = saved[1] if ctx.has_c else None
            gz = gz.contiguous() if gz is not None else None
            gJ = gJ.contiguous() if gJ is not None else None
            need_gc = ctx.has_c and ctx.needs_input_grad[3]
            if not ctx.direct:
                gx, gc, gflat = engine.backward(x, tape, c, gz, gJ, need_gc)
                return (None, None, gx, gc, *(engine.split_flat(gflat) if ctx.n_params else ()))
            out = []

            def run(g_params, accumulate):
                out.append(engine.backward(x, tape, c, gz, gJ, need_gc, g_params=g_params, accumulate=bool(accumulate)))

            flat, views = engine.grad_views()

            def fresh():
                buf = torch.empty(engine.total, dtype=torch.float32, device=x.device)
                return buf, engine.split_flat(buf)

            deliver_param_grads(engine.params, views, flat, run, fresh)
            gx, gc, _ = out[0]
            return None, None, gx, gc, None


def _as_f32_2d(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise HintAmdError(f"hint_amd: {what} is on {t.device}; the coupling block runs on MI355X only "
                           "(no CPU fallback)")
    if t.dtype != torch.float32:
        raise HintAmdError(f"hint_amd: {what} has dtype {t.dtype}; kernels are fp32 only")
    if t.dim() != 2:
        raise HintAmdError(f"hint_amd: {what} must be [batch, features], got {tuple(t.shape)}")
    return t.contiguous()


class NodePermutation(nn.Module):
    """Stand-in for FrEIA's `HouseholderPerm(fixed=True)` of reshuffle=True trees (hint.py:36-39): a
    fixed random orthogonal [D,D] matrix applied to the node's lanes on entry (x W) and undone on
    exit of the inverse pass (x W^T).  FrEIA's own construction is not available, so the matrix is
    drawn here (QR of a Gaussian, torch's global RNG) and stored as a buffer; any orthogonal matrix
    loaded into `W` is used as it is.  Its log-det is 0."""

    def __init__(self, D: int):
        super().__init__()
        q, r = torch.linalg.qr(torch.randn(D, D, dtype=torch.float64))
        self.register_buffer("W", (q * torch.sign(torch.diagonal(r))).to(torch.float32).contiguous())

    def forward(self, x, c=[], rev=False):
        return [x[0] @ (self.W.t() if rev else self.W)]

    def jacobian(self, x, c=[], rev=False):
        return 0.


class HierarchicalAffineCouplingTree(nn.Module):
    """Recursive coupling tree (hint.py:21-101).  Parameter container with the reference's
    layout; `forward` runs the whole (sub)tree in one fused kernel launch."""

    def __init__(self, data_shape, dims_c, conv=False, subnet_constructor=None, c_internal=[], clamp=2,
                 max_splits=-1, min_split_size=2, reshuffle=False, _split_idx=None):
        super().__init__()
        if conv:
            raise NotImplementedError("conv=True (hint.py:15-18,29) is out of scope")
        if len(tuple(data_shape)) != 1:
            raise NotImplementedError("only flat [B, d] data is supported")
        self.data_shape = tuple(data_shape)
        self.clamp = clamp
        widths = list(c_internal)          # the reference mutates the caller's list here (hint.py:33-34);
        if len(widths) == 0:               # we compute the same widths without that side effect
            widths = [self.data_shape[0]]
        if len(widths) == 1:
            widths = widths + widths
        D = self.data_shape[0]
        self.perm = NodePermutation(D) if reshuffle else None          # hint.py:36-39
        # hint.py:41; _split_idx (not a reference keyword) lets the conditional-lane couplings of
        # hint_amd/conditional.py reuse the node kernels with another upper/lower split
        self.split_idx = D // 2 if _split_idx is None else int(_split_idx)
        self.conditional = len(dims_c) > 0
        self.condition_length = sum(dims_c[i][0] for i in range(len(dims_c)))
        # hint.py:27-32,44-45: a custom `subnet_constructor(c_in, c_out, c_internal)` is called as the reference calls it; the
        # kernels implement ONE subnet - Linear-ReLU-Linear-ReLU-Linear with one hidden width (hint.py:10-13) - so whatever it
        # returns must have exactly that structure (its initialisation, names inside the Sequential and dtype handling are its
        # own); anything else raises
        make = linear_subnet_constructor if subnet_constructor is None else subnet_constructor
        c_in, c_out = self.split_idx + self.condition_length, D - self.split_idx
        self.s = _checked_subnet(make(c_in, c_out, widths[0]), c_in, c_out)
        self.t = _checked_subnet(make(c_in, c_out, widths[0]), c_in, c_out)
        if self.s[0].out_features != self.t[0].out_features:
            raise NotImplementedError("the s and t subnets of a node must have the same hidden width")
        if D >= 2 * min_split_size and max_splits != 0 and _split_idx is None:   # hint.py:47
            self.leaf = False
            self.upper = HierarchicalAffineCouplingTree((self.split_idx,), dims_c, conv, subnet_constructor,
                                                        widths[1:], clamp, max_splits - 1, min_split_size, reshuffle)
            self.lower = HierarchicalAffineCouplingTree((D - self.split_idx,), dims_c, conv, subnet_constructor,
                                                        widths[1:], clamp, max_splits - 1, min_split_size, reshuffle)
        else:
            self.leaf = True
        self._engine: Optional[_Engine] = None

    # the engine (plan handle, device arenas: raw pointers) belongs to this object on its device: copies and
    # pickles of the module (copy.deepcopy for an EMA / best-checkpoint copy, torch.save(model)) leave it behind
    # and build their own on first use
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        eng, self._engine = self._engine, None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._engine = eng
        return new

    # same closed forms as hint.py:56-60, kept for API parity (not used by the kernels)
    def e(self, s):
        return torch.exp(self.clamp * 0.636 * torch.atan(s))

    def log_e(self, s):
        return self.clamp * 0.636 * torch.atan(s)

    def _flat_nodes(self, off=0, depth=0):
        out = [(self, off, depth)]
        if not self.leaf:
            out += self.upper._flat_nodes(off, depth + 1)
            out += self.lower._flat_nodes(off + self.split_idx, depth + 1)
        return out

    def engine(self, device) -> _Engine:
        device = torch.device(device)
        if device.type != "cuda":
            raise HintAmdError("hint_amd: no CPU implementation; move the model and data to the GPU")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._engine is None or self._engine.device != device:
            ext = self._engine._ext_arena if self._engine is not None else None
            self._engine = _Engine(self, device)
            if ext is not None and ext.device == device:
                self._engine.bind_external_arena(ext)
        return self._engine

    def forward(self, x, c=[], rev=False):
        """-> (x_out, J) like hint.py:62-101."""
        x = _as_f32_2d(x, "x")
        if x.shape[1] != self.data_shape[0]:
            raise HintAmdError(f"expected {self.data_shape[0]} lanes, got {x.shape[1]}")
        cc = None
        if self.conditional:
            cs = [_as_f32_2d(t, "condition") for t in c]
            cc = cs[0] if len(cs) == 1 else torch.cat(cs, dim=1)
            if cc.shape[1] != self.condition_length or cc.shape[0] != x.shape[0]:
                raise HintAmdError("condition shape %s does not match dims_c / batch" % (tuple(cc.shape),))
        with _Region():
            eng = self._engine
            if eng is None or eng.device != x.device:
                eng = self.engine(x.device)
            eng.ensure_arena()
            eng.pack()
            if not torch.is_grad_enabled():
                return eng.apply(x, cc, rev=rev)
            req = [p.requires_grad for p in eng.params]
            n_req = sum(req)
            if n_req == 0 and not (x.requires_grad or (cc is not None and cc.requires_grad)):
                return eng.apply(x, cc, rev=rev)
            fn = _RevCouplingFn if rev else _CouplingFn
            # 'direct' needs every parameter trainable (a partly frozen block takes the autograd route, which returns
            # gradients for exactly the inputs that ask for one)
            if _PARAM_GRADS == "direct" and n_req == len(req):
                return fn.apply(eng, True, x, cc, eng.anchor())
            if n_req == 0:
                return fn.apply(eng, False, x, cc)
            return fn.apply(eng, False, x, cc, *eng.params)


class HierarchicalAffineCouplingBlock(nn.Module):
    """FrEIA-protocol wrapper (hint.py:104-133)."""

    def __init__(self, dims_in, dims_c=[], conv=False, subnet_constructor=None, c_internal=[], clamp=4.,
                 max_splits=-1, min_split_size=2, reshuffle=False):
        super().__init__()
        assert all([tuple(dims_c[i][1:]) == tuple(dims_in[0][1:]) for i in range(len(dims_c))]), \
            "Dimensions of input and one or more conditions don't agree."
        self.tree = HierarchicalAffineCouplingTree(tuple(dims_in[0]), dims_c=dims_c, conv=conv,
                                                   subnet_constructor=subnet_constructor, c_internal=c_internal,
                                                   clamp=clamp, max_splits=max_splits,
                                                   min_split_size=min_split_size, reshuffle=reshuffle)
        self.jac = None

    def forward(self, x, c=[], rev=False):
        x, self.jac = self.tree.forward(x[0], c, rev=rev)
        return [x]

    def jacobian(self, x, c=[], rev=False):
        return self.jac

    def output_dims(self, input_dims):
        assert len(input_dims) == 1, "Can only use one input."
        return input_dims
