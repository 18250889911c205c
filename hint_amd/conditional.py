"""Conditional two-lane HINT: the modules the reference's `conditional_hint_*_full`
configs take from FrEIA around the recursive block, and the two-lane container.

/root/reference/configs/plus_shape/conditional_hint_4_full.py:55-95 builds

    y lane:  y -> [perm_y_i (i>0)] ---------------------------> AffineCoupling ac_y_i -> ... -> z_y
    x lane:  x -> [perm_x_i (i>0)] -> HierarchicalAffineCouplingBlock hac_x_i
                                   -> ExternalAffineCoupling ac_y_to_x_i (conditions = y lane before ac_y_i) -> ... -> z_x

FrEIA (`AffineCoupling`, `ExternalAffineCoupling`, `F_fully_connected`, `HouseholderPerm`,
`ReversibleGraphNet`) is neither vendored in the reference nor installed, and no reference test
covers it: **parity unpinned** for these modules.  They are implemented here from their
definitions in the paper (soft-clamped affine couplings, hint.py:56-60 uses the same clamp) ON
THE SAME HIP NODE KERNELS as the recursive block: an `AffineCoupling` is a one-node tree
(max_splits = 0), an `ExternalAffineCoupling` a one-node tree whose upper half is empty (all
lanes are transformed, s and t see the condition only).  The subnets are therefore the block's
own Linear-ReLU-Linear-ReLU-Linear s and t nets with `internal_size` hidden units, not FrEIA's
`F_fully_connected`; `F_class` / `F_args` are accepted for call compatibility.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .flow import FixedOrthogonal
from .hint import HierarchicalAffineCouplingBlock, HierarchicalAffineCouplingTree, HintAmdError


class F_fully_connected:          # marker for F_class=...; see the module docstring
    pass


def _internal_size(F_args, default):
    return int((F_args or {}).get("internal_size", default))


class _OneNodeCoupling(nn.Module):
    """FrEIA-protocol wrapper around a single coupling node run by the HIP block kernels"""

    def __init__(self, tree: HierarchicalAffineCouplingTree):
        super().__init__()
        self.tree = tree
        self.jac = None

    def forward(self, x, c=[], rev=False):
        x, self.jac = self.tree.forward(x[0], c, rev=rev)
        return [x]

    def jacobian(self, x, c=[], rev=False):
        return self.jac

    def output_dims(self, input_dims):
        assert len(input_dims) == 1, "Can only use one input."
        return input_dims


class AffineCoupling(_OneNodeCoupling):
    """y' = [y_upper, e(s(y_upper)) * y_lower + t(y_upper)]  (conditional_hint_4_full.py:84-88)"""

    def __init__(self, dims_in, dims_c=[], F_class=F_fully_connected, F_args=None, clamp=5.0):
        D = dims_in[0][0]
        h = _internal_size(F_args, 2 * D)
        super().__init__(HierarchicalAffineCouplingTree((D,), dims_c=list(dims_c), c_internal=[h], clamp=clamp,
                                                        max_splits=0))


class ExternalAffineCoupling(_OneNodeCoupling):
    """x' = e(s(c)) * x + t(c): every lane is transformed, s and t see the condition only
    (conditional_hint_4_full.py:76-82, conditions = the y lane)"""

    def __init__(self, dims_in, dims_c=[], F_class=F_fully_connected, F_args=None, clamp=5.0):
        if len(dims_c) == 0:
            raise HintAmdError("ExternalAffineCoupling needs a condition")
        D = dims_in[0][0]
        h = _internal_size(F_args, 2 * D)
        super().__init__(HierarchicalAffineCouplingTree((D,), dims_c=list(dims_c), c_internal=[h], clamp=clamp,
                                                        max_splits=0, _split_idx=0))


class ConditionalHintFlow(nn.Module):
    """The two-lane graph of conditional_hint_4_full.py:55-95 and the calls train_conditional.py makes
    on it: `z_y, z_x = model([y, x])`, `model.log_jacobian(run_forward=False)`, `x_jac()` (:50-55),
    `y, x = model([z_y, z_x], rev=True)`."""

    def __init__(self, ndim_x: int, ndim_y: int, n_blocks: int, hidden: int, clamp: float = 4.0,
                 perm_seed: int = 1):
        super().__init__()
        self.ndim_x, self.ndim_y, self.n_blocks = ndim_x, ndim_y, n_blocks
        self.perm_y, self.perm_x = nn.ModuleList(), nn.ModuleList()
        self.hac_x, self.ac_y_to_x, self.ac_y = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(n_blocks):
            self.perm_y.append(FixedOrthogonal([(ndim_y,)], seed=perm_seed + 2 * i) if i > 0 else nn.Identity())
            self.perm_x.append(FixedOrthogonal([(ndim_x,)], seed=perm_seed + 2 * i + 1) if i > 0 else nn.Identity())
            self.hac_x.append(HierarchicalAffineCouplingBlock(
                [(ndim_x,)], c_internal=[hidden, hidden // 2, hidden // 4], clamp=clamp))              # :72-75
            self.ac_y_to_x.append(ExternalAffineCoupling([(ndim_x,)], dims_c=[(ndim_y,)],
                                                         F_args={"internal_size": hidden}))             # :78-82
            self.ac_y.append(AffineCoupling([(ndim_y,)], F_args={"internal_size": hidden // 2}))        # :84-88
        self._jac_x = self._jac_y = None

    def forward(self, inputs: Sequence[torch.Tensor], rev: bool = False):
        a, b = inputs
        if not rev:
            y, x = a, b
            jx, jy = 0, 0
            for i in range(self.n_blocks):
                if i > 0:
                    (y,) = self.perm_y[i]([y])
                    (x,) = self.perm_x[i]([x])
                (x,) = self.hac_x[i]([x]);               jx = jx + self.hac_x[i].jacobian(None)
                (x,) = self.ac_y_to_x[i]([x], c=[y]);    jx = jx + self.ac_y_to_x[i].jacobian(None)
                (y,) = self.ac_y[i]([y]);                jy = jy + self.ac_y[i].jacobian(None)
            self._jac_x, self._jac_y = jx, jy
            return y, x
        # inverse: the y lane first (its intermediate values are the x lane's conditions)
        zy, zx = a, b
        conds: List[Optional[torch.Tensor]] = [None] * self.n_blocks
        jx, jy = 0, 0
        y = zy
        for i in reversed(range(self.n_blocks)):
            (y,) = self.ac_y[i]([y], rev=True);          jy = jy + self.ac_y[i].jacobian(None)
            conds[i] = y                                  # y lane as ac_y_to_x_i saw it
            if i > 0:
                (y,) = self.perm_y[i]([y], rev=True)
        x = zx
        for i in reversed(range(self.n_blocks)):
            (x,) = self.ac_y_to_x[i]([x], c=[conds[i]], rev=True);  jx = jx + self.ac_y_to_x[i].jacobian(None)
            (x,) = self.hac_x[i]([x], rev=True);                     jx = jx + self.hac_x[i].jacobian(None)
            if i > 0:
                (x,) = self.perm_x[i]([x], rev=True)
        self._jac_x, self._jac_y = jx, jy
        return y, x

    def log_jacobian(self, inputs=None, rev=False, run_forward=True):
        if run_forward:
            self.forward(inputs, rev=rev)
        return self._jac_x + self._jac_y

    def x_jac(self):
        """train_conditional.py:50-55: log-det of the x lane (hac_x_* and ac_y_to_x_* nodes)"""
        return self._jac_x


class _LossPair:
    """what step() returns on every path: unpacks to the two loss terms as DEVICE SCALARS (torch tensors: `l0 + l1`,
    `sum(batch_losses)`, `torch.stack`, `.item()` all work), evaluated when it is unpacked so that no torch kernel runs inside the
    step or its captured graph.  The sums live in a buffer the next step's prologue clears: unpack before stepping again
    (FlowTrainer.step returns the same type)."""

    def __init__(self, trainer):
        self._t = trainer

    def __iter__(self):
        return iter(self._t.last_losses())


class ConditionalFlowTrainer:
    """Fast training step for a ConditionalHintFlow (what FlowTrainer is for a HintFlow): one
    iteration of /root/reference/train_conditional.py:120-150 -

        x += 0.01*randn_like(x); z_y, z_x = model([y, x]); z = cat(z_x, z_y)
        loss = 0.5*sum(z^2,1).mean() - model.log_jacobian(...).mean()
        loss.backward(); clamp grads to +-5; Adam.step()

    - without an autograd graph and without per-parameter kernels: every module's forward and
    backward is a direct C-ABI launch into model-wide flat parameter / gradient / Adam arenas
    (the 3 500 parameter tensors of the cfg-4 model cost the module path ~65 ms of per-tensor
    launches per step), one re-pack launch, one gradient all-reduce (hint_amd/dp.py), one fused
    clamp+Adam launch.  The two lanes meet in the backward pass: dL/dy of block i is the sum of
    what ac_y_i and ac_y_to_x_i (through its condition) send back.
    use_graph (one process): the whole iteration is captured once per batch shape into a hipGraph and replayed
    (the step counter and Adam's bias corrections live in device memory, written by the re-pack
    launch's prologue, so nothing in the graph depends on the host's step count).

    Round 5 - 25 launches per iteration, none of them torch's (HINT_COND_LEGACY=1: the ~65 of round 4):
      * the y lane is a real chain - ac_y_0, [perm_y_1] ac_y_1, ... depend on y only - and runs as ONE forward launch
        (hint_chain_forward: the 4 x 4 permutations fused; the permuted y the x lane takes as its condition is the top slice
        of that block's tape) and ONE row-parallel backward launch, into which the gradients the x lane sends back through
        its conditions enter per block (ChainBlock::g_add, hint_chain_set_block_io): the x lane's backward runs first;
      * the x lane's eight modules are launches of their own (they are a graph, not a chain: every one needs the y lane),
        forward (hint_block_forward_noisy: the dequantisation noise drawn in the first one) and row-parallel backward
        (hint_block_backward_rows);
      * the weight gradients of the modules of one plan - hac_x x 4, ac_y_to_x x 4, ac_y x 4 - are ONE part B and ONE slab
        reduction per plan, with the clamp + Adam step in the reduction (hint_chain_wgrad_adam: 3 + 3 launches for 12 + 12 +
        the optimizer's): the modules are gathered in chains whose blocks carry their own input and condition pointers;
      * every intermediate lives in a buffer allocated once per batch size, so the chains' tables are built once."""

    def __init__(self, flow: ConditionalHintFlow, lr: float = 0.01 * 3e-2, betas=(0.9, 0.95), eps: float = 1e-4,
                 weight_decay: float = 1.86e-5, grad_clamp: float = 5.0, noise: float = 0.01, group=None,
                 use_graph: bool = True, seed: Optional[int] = None):
        from . import _lib, dp
        self._lib, self._dp = _lib, dp
        self.lib = _lib.load()
        self.flow, self.group = flow, group
        self._lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.grad_clamp, self.noise = grad_clamp, noise
        self.step_count = 0
        self.use_graph = use_graph
        self._graph, self._static, self._out = None, None, None
        dev = next(flow.parameters()).device
        if dev.type != "cuda":
            raise HintAmdError("ConditionalFlowTrainer needs the model on a GPU (no CPU path)")
        self.device = dev
        self.mods = []                                    # (kind, block index, module) in forward order
        for i in range(flow.n_blocks):
            self.mods += [("hac_x", i, flow.hac_x[i]), ("ac_y_to_x", i, flow.ac_y_to_x[i]), ("ac_y", i, flow.ac_y[i])]
        self.engines = [m.tree.engine(dev) for _, _, m in self.mods]
        self.slices, cursor = [], 0
        for e in self.engines:
            self.slices.append((cursor, cursor + e.total))
            cursor += e.total
        self.n_floats = cursor
        self.P, self.G, self.M, self.V = (torch.zeros(cursor, dtype=torch.float32, device=dev) for _ in range(4))
        for e, (a, b) in zip(self.engines, self.slices):
            e.bind_external_arena(self.P[a:b])
            e.ensure_arena()
            e.pack()
        if dp.world_info(group)[1] > 1:
            # data-parallel replicas start from rank 0's weights AND buffers (the fixed permutations are drawn per process:
            # replicas with different matrices would sum gradients of different functions) - as FlowTrainer does
            src = torch.distributed.get_global_rank(group, 0) if group is not None else 0
            torch.distributed.broadcast(self.P, src=src, group=group)
            for buf in flow.buffers():
                if buf.is_cuda and buf.numel() > 0:
                    torch.distributed.broadcast(buf, src=src, group=group)
            for e in self.engines:
                e._perm_key = None          # (composed permutations are rebuilt from the received matrices)
                e.pack()
        self._pack_group, self._pack_key = None, None
        self.last = None
        # device-side step state (see FlowTrainer): opt_state = {lr, beta1, beta2, lr/(1-beta1^t),
        # 1/sqrt(1-beta2^t), ...}, rng_state[1] = step count; the re-pack launch's prologue advances them
        self.opt_state = torch.tensor([lr, betas[0], betas[1], 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        if seed is None:            # (not from torch's global generator: building a trainer must not shift the caller's random stream)
            import os
            seed = int.from_bytes(os.urandom(8), "little") >> 2
        rank = dp.world_info(group)[0]
        self.rng_state = torch.tensor([(seed + 0x9E3779B97F4A7C15 * rank) & (2 ** 63 - 1), 0], dtype=torch.int64, device=dev)
        self._st = {}                # per batch size: static buffers and chains (_state_for)
        import os
        self._legacy = os.environ.get("HINT_COND_LEGACY", "0") not in ("", "0")
        self.loss_acc = torch.zeros(64, 2, dtype=torch.float32, device=dev)     # hint_block_forward_ex: the two loss sums

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, v: float):
        self._lr = float(v)
        self.opt_state[0] = self._lr            # read by the optimizer launch on the device: no re-capture

    def __del__(self):
        try:
            self._graph = None
            for st in getattr(self, "_st", {}).values():
                for h in st["chains"].values():
                    self.lib.hint_chain_destroy(h)
            self._st = {}
            if getattr(self, "_pack_group", None):
                self.lib.hint_pack_group_destroy(self._pack_group)
        except Exception:
            pass

    # ---- static buffers and chains of a batch size ----------------------------------------------------
    KINDS = ("hac_x", "ac_y_to_x", "ac_y")

    def _state_for(self, B: int):
        import ctypes as C
        flow, dev, nb = self.flow, self.device, self.flow.n_blocks
        eng = dict(zip([(k, i) for k, i, _ in self.mods], self.engines))
        sl = dict(zip([(k, i) for k, i, _ in self.mods], self.slices))
        key = tuple((e.arena.data_ptr(), e.packed.data_ptr()) for e in self.engines) + (self.G.data_ptr(),)
        st = self._st.get(B)
        if st is not None and st["key"] == key:
            return st
        if st is not None:                     # an arena or packed buffer moved: graphs and chains point at the old addresses
            self._graph = None
            torch.cuda.synchronize(dev)        # (launches that read the chains' device tables may still be queued)
            for h in st["chains"].values():
                self.lib.hint_chain_destroy(h)
            del self._st[B]
        if len(self._st) >= 4:                 # ragged data: forget the batch sizes no live graph was captured on
            keep = self._static[0].shape[0] if (self._graph is not None and self._static is not None) else None
            torch.cuda.synchronize(dev)
            for b in [b for b in self._st if b != keep]:
                for h in self._st[b]["chains"].values():
                    self.lib.hint_chain_destroy(h)
                del self._st[b]
        dx, dy = flow.ndim_x, flow.ndim_y
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        st = {"key": key, "B": B}
        st["xn"] = f32(B, dx)
        st["xa"] = [f32(B, dx) for _ in range(nb)]; st["xb"] = [f32(B, dx) for _ in range(nb)]
        st["Jxa"] = [f32(B) for _ in range(nb)]; st["Jxb"] = [f32(B) for _ in range(nb)]
        st["zy"], st["Jy"], st["gy"] = f32(B, dy), f32(B), f32(B, dy)
        st["gxa"] = [f32(B, dx) for _ in range(nb)]; st["gxb"] = [f32(B, dx) for _ in range(nb)]
        st["gc"] = [f32(B, dy) for _ in range(nb)]
        st["x_in"], st["y_in"] = f32(B, dx), f32(B, dy)          # the step's inputs (a captured graph reads them here)
        # fixed matrices in front of the modules (composed with the modules' own node permutations: none here)
        st["perm"] = {("hac_x", i): eng[("hac_x", i)].compose_perm(flow.perm_x[i].W if i > 0 else None) for i in range(nb)}
        st["perm"].update({("ac_y", i): eng[("ac_y", i)].compose_perm(flow.perm_y[i].W if i > 0 else None) for i in range(nb)})
        st["perm"].update({("ac_y_to_x", i): eng[("ac_y_to_x", i)].compose_perm(None) for i in range(nb)})
        st["tapes"], st["ws"], st["wsb"], st["chains"] = {}, {}, {}, {}
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            for kind in self.KINDS:
                e0 = eng[(kind, 0)]
                if any(eng[(kind, i)].shape_key != e0.shape_key for i in range(nb)):
                    raise HintAmdError("ConditionalFlowTrainer: the modules of one kind must share a plan")
                tf = max(self.lib.hint_plan_tape_floats(e0.plan, B), 1)
                wb = (self.lib.hint_plan_workspace_bytes(e0.plan, B) + 255) // 256 * 256
                st["tapes"][kind] = torch.empty(nb, tf, dtype=torch.float32, device=dev)
                st["ws"][kind] = torch.empty(nb, max(wb, 256), dtype=torch.uint8, device=dev)
                st["wsb"][kind] = wb
                h = C.c_void_p()
                self._lib.check(self.lib.hint_chain_create(e0.plan, nb, B, C.byref(h)), "hint_chain_create")
                st["chains"][kind] = h
                for i in range(nb):
                    e = eng[(kind, i)]
                    a, _ = sl[(kind, i)]
                    self._lib.check(self.lib.hint_chain_set_block(
                        h, i, e.arena.data_ptr(), e.packed.data_ptr(), ptr(st["perm"][(kind, i)]), st["tapes"][kind][i].data_ptr(),
                        st["ws"][kind][i].data_ptr(), wb, self.G.data_ptr() + 4 * a), "hint_chain_set_block")
            # the y lane after its permutation = the x lane's condition: block i's tape TOP slice - the kernels keep a block's permuted
            # input at tape + (n_levels - 1) * B * d (hint_fwd.hip store_tile, hint_wgrad.hip wsrc); block 0 has no permutation in
            # front: the input itself
            top = max(dep for _, _, dep in flow.ac_y[0].tree._flat_nodes()) * B * dy          # (n_levels - 1) slices down
            st["yp"] = [st["y_in"]] + [st["tapes"]["ac_y"][i][top:top + B * dy].view(B, dy) for i in range(1, nb)]
            st["hx0_in"] = st["xn"]
            for i in range(nb):
                self._lib.check(self.lib.hint_chain_set_block_io(st["chains"]["ac_y"], i, None, None, st["gc"][i].data_ptr()),
                                "hint_chain_set_block_io")
                hx_in = None if st["perm"][("hac_x", i)] is not None else (st["xn"] if i == 0 else st["xb"][i - 1])     # (block 0: _iteration keeps it in step with `noise`)
                self._lib.check(self.lib.hint_chain_set_block_io(st["chains"]["hac_x"], i, ptr(hx_in), None, None),
                                "hint_chain_set_block_io")
                self._lib.check(self.lib.hint_chain_set_block_io(st["chains"]["ac_y_to_x"], i, st["xa"][i].data_ptr(),
                                                                 st["yp"][i].data_ptr(), None), "hint_chain_set_block_io")
            for kind in self.KINDS:
                self._lib.check(self.lib.hint_chain_commit(st["chains"][kind]), "hint_chain_commit")
        self._st[B] = st
        return st

    def _pack_all(self, prologue: bool = False):
        import ctypes as C
        key = tuple((e.arena.data_ptr(), e.packed.data_ptr()) for e in self.engines)
        if self._pack_key != key:
            if self._pack_group:
                self.lib.hint_pack_group_destroy(self._pack_group)
            n = len(self.engines)
            plans = (C.c_void_p * n)(*[e.plan.value for e in self.engines])
            params = (C.c_void_p * n)(*[e.arena.data_ptr() for e in self.engines])
            packed = (C.c_void_p * n)(*[e.packed.data_ptr() for e in self.engines])
            handle = C.c_void_p()
            with torch.cuda.device(self.device):
                self._lib.check(self.lib.hint_pack_group_create(plans, params, packed, n, C.byref(handle)),
                                "hint_pack_group_create")
            self._pack_group, self._pack_key = handle, key
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            if prologue:      # also: step counter += 1, Adam's bias corrections of that step -> opt_state
                st = self.lib.hint_pack_group_run_ex(self._pack_group, self.loss_acc.data_ptr(), self.loss_acc.numel(),
                                                     self.rng_state.data_ptr(), self.opt_state.data_ptr(), stream)
            else:
                st = self.lib.hint_pack_group_run(self._pack_group, stream)
        self._lib.check(st, "hint_pack_group_run")

    def allreduce_plan(self) -> str:
        """what the step does with the gradient arena between backward and optimizer (for bench.py's config line)"""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return "none (one process)"
        world = self._dp.world_info(self.group)[1]
        where = "captured in the step's hipGraph" if self._graph is not None else "issued from the host"
        return (f"one all-reduce (sum) of the flat fp32 gradient arena ({self.n_floats} floats) over {world} ranks "
                f"({torch.distributed.get_backend(self.group)}); {where}")

    def _graphable(self) -> bool:
        """one process, or a data-parallel job over RCCL (whose all-reduce is captured with the step)"""
        if not self.use_graph or getattr(self, "_graph_failed", False):
            return False
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return True
        import os
        return torch.distributed.get_backend(self.group) == "nccl" and os.environ.get("HINT_GRAPH_ALLREDUCE", "1") != "0"

    def step(self, x: torch.Tensor, y: torch.Tensor):
        """one iteration on this rank's rows; returns a pair that unpacks to the device scalars (0.5*|z|^2 mean, -log|det J| mean)
        - `l0, l1 = trainer.step(x, y)`; unpack it before the next step (its prologue clears the sums)"""
        for e in self.engines:
            e.ensure_arena()
        if not self._graphable():
            return self._iteration(x, y, on_device_adam=False)
        if self._graph is not None and not self._legacy:
            # an arena or packed buffer that moved after the capture (ensure_arena above re-gathers): the graph holds the old addresses
            st = self._st.get(self._static[0].shape[0])
            if st is None or st["key"] != tuple((e.arena.data_ptr(), e.packed.data_ptr()) for e in self.engines) + (self.G.data_ptr(),):
                self._graph = None
        if self._graph is None or self._static[0].shape != x.shape or self._static[1].shape != y.shape:
            if not self._capture(x, y):
                return self._iteration(x, y, on_device_adam=False)
        if x.data_ptr() != self._static[0].data_ptr():
            self._static[0].copy_(x)
        if y.data_ptr() != self._static[1].data_ptr():
            self._static[1].copy_(y)
        self._graph.replay()
        self.step_count += 1
        return self._out

    def input_buffers(self, x: torch.Tensor, y: torch.Tensor):
        """the captured step's own input tensors holding a copy of the arguments (FlowTrainer.input_buffers)"""
        if not self._graphable():
            return x, y
        for e in self.engines:
            e.ensure_arena()
        if self._graph is None or self._static[0].shape != x.shape or self._static[1].shape != y.shape:
            if not self._capture(x, y):
                return x, y
        self._static[0].copy_(x); self._static[1].copy_(y)
        return self._static

    def _capture(self, x, y):
        if self._legacy:
            sx, sy = x.clone(), y.clone()
        else:                                  # the fast path reads its inputs from the batch size's static buffers
            st = self._state_for(x.shape[0])
            sx, sy = st["x_in"], st["y_in"]
            sx.copy_(x); sy.copy_(y)
        snap = [t.clone() for t in (self.P, self.M, self.V)]
        state = (self.opt_state.clone(), self.rng_state.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # warm-up on a side stream (allocator, kernel loading) ...
            self._iteration(sx, sy, on_device_adam=True)
        torch.cuda.current_stream(self.device).wait_stream(side)
        for t, s0 in zip((self.P, self.M, self.V), snap):      # ... whose optimizer step is taken back
            t.copy_(s0)
        self.opt_state.copy_(state[0]); self.rng_state.copy_(state[1])
        self.G.zero_()
        self.rng_state[1] = self.step_count
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local" if dist_on else "global"):
                self._out = self._iteration(sx, sy, on_device_adam=True)
        except Exception:
            if not dist_on:
                raise
            self._graph_failed = True           # the collective could not be captured: plain launches from now on
            torch.cuda.synchronize(self.device)
            for t, s0 in zip((self.P, self.M, self.V), snap):
                t.copy_(s0)
            self.opt_state.copy_(state[0]); self.rng_state.copy_(state[1])
            self.G.zero_()
            return False
        self._graph, self._static = g, (sx, sy)
        return True

    def _iteration(self, x: torch.Tensor, y: torch.Tensor, on_device_adam: bool):
        """One iteration as 25 launches (class docstring): re-pack; y chain forward; per block hac_x, ac_y_to_x forward; per
        block (last first) ac_y_to_x, hac_x row-parallel backward; y chain row-parallel backward (the conditions' gradients
        enter per block); per plan part B + slab reduction with clamp + Adam in it.  Data-parallel jobs and the host-stepped
        optimizer keep part B, the all-reduce and the optimizer launch apart."""
        if self._legacy:
            return self._iteration_legacy(x, y, on_device_adam)
        flow, B, lib, chk = self.flow, x.shape[0], self.lib, self._lib.check
        if B == 0:
            raise HintAmdError("ConditionalFlowTrainer: empty batch")
        st = self._state_for(B)
        if x.data_ptr() != st["x_in"].data_ptr():
            st["x_in"].copy_(x)
        if y.data_ptr() != st["y_in"].data_ptr():
            st["y_in"].copy_(y)
        self._pack_all(prologue=on_device_adam)         # (its prologue clears loss_acc and advances the step / noise counter)
        if not on_device_adam:
            self.loss_acc.zero_()
            self.rng_state[1] += 1
        eng = dict(zip([(k, i) for k, i, _ in self.mods], self.engines))
        nb = flow.n_blocks
        ptr = lambda t: t.data_ptr() if t is not None else None
        noisy = self.noise > 0
        x0 = st["xn"] if noisy else st["x_in"]           # what hac_x_0 saw
        if st.get("hx0_in") is not x0:                   # (hac_x_0's own input pointer in its gathered chain follows `noise`)
            if st["perm"][("hac_x", 0)] is None:
                if torch.cuda.is_current_stream_capturing():
                    raise HintAmdError("ConditionalFlowTrainer: `noise` switched on / off: call step() once outside a capture")
                # (hint_chain_commit copies the table with a synchronous hipMemcpy on the null stream: the launches of the step before
                #  - possibly on a non-blocking stream - must be through with the old one)
                torch.cuda.current_stream(self.device).synchronize()
                with torch.cuda.device(self.device):
                    chk(lib.hint_chain_set_block_io(st["chains"]["hac_x"], 0, x0.data_ptr(), None, None), "hint_chain_set_block_io")
                    chk(lib.hint_chain_commit(st["chains"]["hac_x"]), "hint_chain_commit")
            st["hx0_in"] = x0
        tape = lambda kind, i: st["tapes"][kind][i].data_ptr()
        ws = lambda kind, i: st["ws"][kind][i].data_ptr()
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        fuse = on_device_adam and not dist_on
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            # ---- forward: the y lane (one launch), then the x lane's modules ----
            chk(lib.hint_chain_forward(st["chains"]["ac_y"], st["y_in"].data_ptr(), None, st["zy"].data_ptr(), st["Jy"].data_ptr(), None,
                                       self.loss_acc.data_ptr(), stream), "hint_chain_forward")
            for i in range(nb):
                e = eng[("hac_x", i)]
                xin = st["x_in"] if i == 0 else st["xb"][i - 1]
                nz = float(self.noise) if (i == 0 and noisy) else 0.0
                chk(lib.hint_block_forward_noisy(e.plan, e.arena.data_ptr(), e.packed.data_ptr(), xin.data_ptr(), None, st["xa"][i].data_ptr(),
                                                 st["Jxa"][i].data_ptr(), tape("hac_x", i), ptr(st["perm"][("hac_x", i)]),
                                                 st["Jxb"][i - 1].data_ptr() if i > 0 else None, None, nz,
                                                 self.rng_state.data_ptr() if nz else None, st["xn"].data_ptr() if nz else None, B, stream),
                    "hint_block_forward_noisy")
                e = eng[("ac_y_to_x", i)]
                chk(lib.hint_block_forward_noisy(e.plan, e.arena.data_ptr(), e.packed.data_ptr(), st["xa"][i].data_ptr(), st["yp"][i].data_ptr(),
                                                 st["xb"][i].data_ptr(), st["Jxb"][i].data_ptr(), tape("ac_y_to_x", i), None,
                                                 st["Jxa"][i].data_ptr(), self.loss_acc.data_ptr() if i == nb - 1 else None, 0.0, None, None,
                                                 B, stream), "hint_block_forward_noisy")
            # ---- backward, row-parallel parts: the x lane (dL/dz = z / B and dL/dJ = -1/B applied on load), then the y chain ----
            for i in reversed(range(nb)):
                e = eng[("ac_y_to_x", i)]
                gz = st["xb"][nb - 1] if i == nb - 1 else st["gxb"][i + 1]
                chk(lib.hint_block_backward_rows(e.plan, e.arena.data_ptr(), e.packed.data_ptr(), st["xa"][i].data_ptr(), tape("ac_y_to_x", i),
                                                 st["yp"][i].data_ptr(), gz.data_ptr(), None, st["gxa"][i].data_ptr(), st["gc"][i].data_ptr(),
                                                 ws("ac_y_to_x", i), st["wsb"]["ac_y_to_x"], None, 1.0 / B if i == nb - 1 else 1.0, -1.0 / B,
                                                 B, stream), "hint_block_backward_rows")
                e = eng[("hac_x", i)]
                xin = x0 if i == 0 else st["xb"][i - 1]
                chk(lib.hint_block_backward_rows(e.plan, e.arena.data_ptr(), e.packed.data_ptr(), xin.data_ptr(), tape("hac_x", i), None,
                                                 st["gxa"][i].data_ptr(), None, st["gxb"][i].data_ptr(), None, ws("hac_x", i),
                                                 st["wsb"]["hac_x"], ptr(st["perm"][("hac_x", i)]), 1.0, -1.0 / B, B, stream),
                    "hint_block_backward_rows")
            chk(lib.hint_chain_backward_parts(st["chains"]["ac_y"], st["y_in"].data_ptr(), None, st["zy"].data_ptr(), None, st["gy"].data_ptr(),
                                              None, 1.0 / B, -1.0 / B, 1, 1, stream), "hint_chain_backward_parts")
            # ---- weight gradients: one part B + slab reduction per plan ----
            for kind in self.KINDS:
                xk = st["y_in"].data_ptr() if kind == "ac_y" else None
                if fuse:
                    chk(lib.hint_chain_wgrad_adam(st["chains"][kind], xk, None, self.P.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                  self.n_floats, self.opt_state.data_ptr(), self.betas[0], self.betas[1], self.eps, self.wd,
                                                  1.0, self.grad_clamp, stream), "hint_chain_wgrad_adam")
                else:
                    chk(lib.hint_chain_wgrad_range(st["chains"][kind], xk, None, 0, 0, nb, stream), "hint_chain_wgrad_range")
        if not fuse:
            scale = self._dp.allreduce_sum_(self.G, self.group)       # (no-op without a process group)
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device).cuda_stream
                if on_device_adam:
                    stt = lib.hint_adam_step_dev(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.n_floats,
                                                 self.opt_state.data_ptr(), self.betas[0], self.betas[1], self.eps, self.wd, scale,
                                                 self.grad_clamp, 1, stream)
                else:
                    self.step_count += 1
                    stt = lib.hint_adam_step(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.n_floats,
                                             self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, scale,
                                             self.grad_clamp, 1, stream)
            chk(stt, "hint_adam_step")
            if not on_device_adam:
                self.rng_state[1] = self.step_count          # keep the device counter in step for a later capture
        self.last = (st["zy"], st["xb"][nb - 1], st["Jxb"][nb - 1], st["Jy"])
        self._last_B = B
        return _LossPair(self)

    def last_losses(self):
        """(0.5 |z|^2 mean, -log|det J| mean) of the most recent step as device scalars, from the launches' loss sums (read
        them before the next step: its re-pack launch clears the sums)"""
        s = self.loss_acc.sum(dim=0)
        return s[0] / self._last_B, -s[1] / self._last_B

    def _iteration_legacy(self, x: torch.Tensor, y: torch.Tensor, on_device_adam: bool):
        """Round 4's iteration (HINT_COND_LEGACY=1; kept for A/B runs): both lanes forward and backward as direct launches.  What FrEIA's graph does between the couplings is folded
        into them (hint_block_*_ex): the x lane's fixed permutation rides in front of hac_x, the log-dets accumulate
        through J_in, the two loss sums come from the last couplings' launches (loss_acc), dL/dz = z / B and
        dL/dJ = -1/B are applied on load.  Left to torch: the y lane's 4 x 4 permutation (the permuted y is also the
        x lane's condition) and the sum of the two gradients that meet in y."""
        flow, B = self.flow, x.shape[0]
        self._pack_all(prologue=on_device_adam)         # (its prologue clears loss_acc)
        if not on_device_adam:
            self.loss_acc.zero_()
        if self.noise > 0:
            x = x.add(torch.randn_like(x), alpha=self.noise)
        eng = dict(zip([(k, i) for k, i, _ in self.mods], self.engines))
        sl = dict(zip([(k, i) for k, i, _ in self.mods], self.slices))
        saved = {}
        nb = flow.n_blocks
        Jx = Jy = None
        for i in range(nb):                                         # ---- forward, both lanes ----
            front = None
            if i > 0:
                y = y @ flow.perm_y[i].W
                front = flow.perm_x[i].W
            acc = self.loss_acc if i == nb - 1 else None

            def fwd(kind, inp, c, front, J_in, acc):
                e = eng[(kind, i)]
                out, J, tape = e.forward_chain(inp, c, e.compose_perm(front), J_in, acc, True)
                saved[(kind, i)] = (inp, tape, c, front)
                return out, J
            x, Jx = fwd("hac_x", x, None, front, Jx, None)
            x, Jx = fwd("ac_y_to_x", x, y, None, Jx, acc)
            y, Jy = fwd("ac_y", y, None, None, Jy, acc)
        zx, zy = x, y
        sums = self.loss_acc.sum(dim=0)
        l0, l1 = sums[0] / B, -sums[1] / B
        gx, gy = zx, zy                                             # dL/dz = z / B: the scale is applied on load
        for i in reversed(range(nb)):                               # ---- backward ----
            scale = 1.0 / B if i == nb - 1 else 1.0

            def bwd(kind, g, need_gc=False, scale=1.0):
                xin, tape, c, front = saved[(kind, i)]
                a, b = sl[(kind, i)]
                gin, gc, _ = eng[(kind, i)].backward(xin, tape, c, g, None, need_gc, self.G[a:b], accumulate=True, front=front,
                                                     gz_scale=scale, gJ_const=-1.0 / B)
                return gin, gc
            gy, _ = bwd("ac_y", gy, scale=scale)
            gx, gc = bwd("ac_y_to_x", gx, need_gc=True, scale=scale)
            gy = gy + gc                                            # the condition of ac_y_to_x_i is the y lane
            gx, _ = bwd("hac_x", gx)                                # (comes back through the x lane's permutation)
            if i > 0:
                gy = gy @ flow.perm_y[i].W.t()
        if on_device_adam:                      # step factors come from opt_state (prologue above): capturable
            scale = self._dp.allreduce_sum_(self.G, self.group)       # (no-op without a process group)
            with torch.cuda.device(self.device):
                st = self.lib.hint_adam_step_dev(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                 self.n_floats, self.opt_state.data_ptr(), self.betas[0], self.betas[1],
                                                 self.eps, self.wd, scale, self.grad_clamp, 1,
                                                 torch.cuda.current_stream(self.device).cuda_stream)
            self._lib.check(st, "hint_adam_step_dev")
        else:
            scale = self._dp.allreduce_sum_(self.G, self.group)
            self.step_count += 1
            with torch.cuda.device(self.device):
                st = self.lib.hint_adam_step(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                             self.n_floats, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                             self.wd, scale, self.grad_clamp, 1,
                                             torch.cuda.current_stream(self.device).cuda_stream)
            self._lib.check(st, "hint_adam_step")
            self.rng_state[1] = self.step_count          # keep the device counter in step for a later capture
        self.last = (zy, zx, Jx, Jy)
        self._last_B = B
        return _LossPair(self)
