"""Minimal sequential flow container: the part of FrEIA's `ReversibleGraphNet` the reference
configs use around the coupling block, so the BASELINE configs run end to end
without FrEIA (which is neither vendored in the reference nor installed here).

Mirrors what /root/reference/configs/uci_data/power_hint_8.py:55-77 builds:
    InputNode -> [HouseholderPerm(fixed=True) (i>0) -> HierarchicalAffineCouplingBlock] x n -> OutputNode
and the three calls the training loop makes on it (train_unconditional.py:124-125,153):
    z = model(x) ; model.log_jacobian(x, run_forward=False) ; model(z, rev=True)
FrEIA's HouseholderPerm arithmetic is unavailable (unpinned third-party), so the fixed
inter-block permutation is a stored random orthogonal matrix (log-det 0), kept in the
state_dict.  "parity unpinned" applies to that stand-in only, not to the coupling blocks.
"""
from __future__ import annotations

import ctypes as C
import operator
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from . import hint as _hint
from .hint import HierarchicalAffineCouplingBlock, HintAmdError, _Lease, _Region, _mark_launch, _NONES


def random_orthogonal(d: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(d, d, generator=g, dtype=torch.float64))
    # QR returns column-major storage; the kernels read W through its raw pointer as row-major
    return (q * torch.sign(torch.diagonal(r))).to(torch.float32).contiguous()


class FixedOrthogonal(nn.Module):
    """x -> x @ W with a fixed orthogonal W; FrEIA-protocol module (list in / list out)."""

    def __init__(self, dims_in, dims_c=[], seed: int = 0, W: Optional[torch.Tensor] = None):
        super().__init__()
        d = dims_in[0][0]
        self.register_buffer("W", random_orthogonal(d, seed) if W is None else W.clone().to(torch.float32).contiguous())

    def forward(self, x, c=[], rev=False):
        return [x[0] @ (self.W.t() if rev else self.W)]

    def jacobian(self, x, c=[], rev=False):
        return 0.0

    def output_dims(self, input_dims):
        return input_dims


_REQUIRES_GRAD = operator.attrgetter("requires_grad")
_GRAD = operator.attrgetter("grad")


class ChainRunner:
    """The blocks of a HintFlow as CHAINED launches (hint_chain_*: every block of the flow in one kernel per pass, the fixed
    permutations fused): the handles, tapes, workspaces and the one-launch re-pack of all blocks.  Shared by HintFlow's fused
    module route (below) and FlowTrainer (hint_amd/train.py).  Parameters stay where the blocks' engines keep them (their own
    arenas or a trainer's model-wide one): a chain is keyed on those addresses and rebuilt when one moves."""

    def __init__(self, flow: "HintFlow", device: torch.device):
        self.lib = _lib.load()
        self.flow, self.device = flow, device
        self.engines = [blk.tree.engine(device) for blk in flow.blocks]
        self.chainable = len(self.engines) > 0 and all(e.shape_key == self.engines[0].shape_key for e in self.engines)
        self.slices, cursor = [], 0
        for e in self.engines:
            self.slices.append((cursor, cursor + e.total))
            cursor += e.total
        self.n_floats = cursor
        self.params = [p for e in self.engines for p in e.params]
        self._pack_group, self._pack_key = None, None
        self._pool = {}             # module route: (B, stream) -> free chain instances (handle, key, keep-alive)
        self._infer = {}            # B -> inference chain (no tape, no workspace)
        self._wsbuf = {}            # (B, stream) -> the backward workspaces (scratch inside one stream-ordered call)
        self.G: Optional[torch.Tensor] = None           # 'direct' parameter gradients of the whole flow ...
        self.gviews: List[torch.Tensor] = []            # ... one cached view per parameter
        self._anchor: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            for pool in self._pool.values():
                for inst in pool:
                    self.lib.hint_chain_destroy(inst[0])
            self._pool = {}
            for handle, _, _ in self._infer.values():
                self.lib.hint_chain_destroy(handle)
            self._infer = {}
            if self._pack_group:
                self.lib.hint_pack_group_destroy(self._pack_group)
                self._pack_group = None
        except Exception:
            pass

    # ---- parameters ------------------------------------------------------------------------------
    def check_arenas(self):
        """parameters rebound from outside (p.data = ..., load_state_dict into new storage, .to()) are pulled back into the
        arenas"""
        for e in self.engines:
            e.ensure_arena()
            if e.packed is None or e.packed.device != self.device:
                e.pack()            # (allocates the packed buffer; pack_all re-packs all of them in one launch)

    def perms(self):
        """permutation in front of block i: the flow's fixed matrix, then the block's own node permutations
        (reshuffle=True trees), composed into one [d,d] matrix"""
        flow = self.flow
        for i in range(flow.n_blocks):          # the kernels read W through its raw pointer (row-major)
            if flow.has_perm(i) and not flow.perms[i].W.is_contiguous():
                flow.perms[i].W = flow.perms[i].W.contiguous()
        front = [flow.perms[i].W if flow.has_perm(i) else None for i in range(flow.n_blocks)]
        return [e.compose_perm(f) for e, f in zip(self.engines, front)]

    def pack_all(self, zero_buf=None, rng_state=None, opt_state=None):
        """one launch re-packs every block (hint_pack_group_*); the group is rebuilt whenever an arena or packed buffer
        moved.  With zero_buf / rng_state / opt_state the launch is a training step's prologue (hint_pack_group_run_ex)."""
        key = tuple((e.arena.data_ptr(), e.packed.data_ptr()) for e in self.engines)
        _mark_launch()
        if self._pack_key != key:
            if self._pack_group:
                self.lib.hint_pack_group_destroy(self._pack_group)
            n = len(self.engines)
            plans = (C.c_void_p * n)(*[e.plan.value for e in self.engines])
            params = (C.c_void_p * n)(*[e.arena.data_ptr() for e in self.engines])
            packed = (C.c_void_p * n)(*[e.packed.data_ptr() for e in self.engines])
            handle = C.c_void_p()
            with torch.cuda.device(self.device):
                _lib.check(self.lib.hint_pack_group_create(plans, params, packed, n, C.byref(handle)),
                           "hint_pack_group_create")
            self._pack_group, self._pack_key = handle, key
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            if zero_buf is not None or rng_state is not None:
                st = self.lib.hint_pack_group_run_ex(self._pack_group, zero_buf.data_ptr() if zero_buf is not None else None,
                                                     zero_buf.numel() if zero_buf is not None else 0,
                                                     rng_state.data_ptr() if rng_state is not None else None,
                                                     opt_state.data_ptr() if opt_state is not None else None, stream)
            else:
                st = self.lib.hint_pack_group_run(self._pack_group, stream)
        _lib.check(st, "hint_pack_group_run")

    # ---- chains ----------------------------------------------------------------------------------
    def chain_key(self, B: int, perms, G: Optional[torch.Tensor]):
        return (B,) + tuple((e.arena.data_ptr(), e.packed.data_ptr()) for e in self.engines) \
            + tuple(p.data_ptr() if p is not None else 0 for p in perms) + ((G.data_ptr(),) if G is not None else ())

    def build_chain(self, B: int, perms, G: Optional[torch.Tensor], ws: Optional[torch.Tensor] = None):
        """-> (handle, keep-alive buffers): a chain for batch size B.  With G (the flat gradient arena, slices in block order)
        it is a training chain - tapes and backward workspaces of all blocks allocated here (ws: share these workspaces) -
        without, an inference chain (hint_chain_inverse and an inference forward touch neither)."""
        e0, n = self.engines[0], len(self.engines)
        tapes = wsb = None
        ws_bytes = 0
        if G is not None:
            tape_floats, ws_bytes = e0.sizes(B)
            ws_bytes = (ws_bytes + 255) // 256 * 256
            tapes = torch.empty(n, tape_floats, dtype=torch.float32, device=self.device)
            wsb = ws if ws is not None else torch.empty(n, ws_bytes, dtype=torch.uint8, device=self.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.hint_chain_create(e0.plan, n, B, C.byref(handle)), "hint_chain_create")
            for i, e in enumerate(self.engines):
                a, _ = self.slices[i]
                _lib.check(self.lib.hint_chain_set_block(
                    handle, i, e.arena.data_ptr(), e.packed.data_ptr(),
                    perms[i].data_ptr() if perms[i] is not None else None,
                    tapes[i].data_ptr() if tapes is not None else None, wsb[i].data_ptr() if wsb is not None else None,
                    ws_bytes, (G.data_ptr() + 4 * a) if G is not None else None), "hint_chain_set_block")
            _lib.check(self.lib.hint_chain_commit(handle), "hint_chain_commit")
        return handle, (tapes, wsb, perms)

    def chain_infer(self, B: int):
        """the chain handle of the sampling / evaluation direction for batch size B: a 100 k-row evaluation batch costs no
        memory beyond its own rows, and asking for a new batch size never evicts a training chain"""
        perms = self.perms()
        key = self.chain_key(B, perms, None)
        have = self._infer.get(B)
        if have is not None and have[1] == key:
            return have[0]
        if have is not None:
            torch.cuda.synchronize(self.device)         # (a launch on the stale chain's device table may still be queued)
            self.lib.hint_chain_destroy(have[0])
            del self._infer[B]
        elif len(self._infer) >= 32:
            torch.cuda.synchronize(self.device)
            for b in list(self._infer):
                self.lib.hint_chain_destroy(self._infer[b][0])
                del self._infer[b]
        handle, keep = self.build_chain(B, perms, None)
        self._infer[B] = (handle, key, keep)
        return handle

    # ---- the module route's training chains: pooled, leased to autograd nodes ---------------------------------
    def grad_views(self):
        if self.G is None or self.G.device != self.device:
            self.G = torch.zeros(max(self.n_floats, 4), dtype=torch.float32, device=self.device)
            self.gviews = [v for e, (a, b) in zip(self.engines, self.slices) for v in e.split_flat(self.G[a:b])]
        return self.G, self.gviews

    def anchor(self) -> torch.Tensor:
        if self._anchor is None:
            self._anchor = torch.zeros(1, dtype=torch.float32, device=self.device, requires_grad=True)
        return self._anchor

    def acquire(self, B: int):
        """-> ((handle, key, keep-alive), lease) for one training forward + backward on the module route.  A chain owns its
        tapes, so a forward that runs while an older autograd node of this flow is still alive takes another instance; the
        lease (kept on the node) hands the instance back when the node dies."""
        G, _ = self.grad_views()
        perms = self.perms()
        key = self.chain_key(B, perms, G)
        skey = (B, torch.cuda.current_stream(self.device).cuda_stream)
        pool = self._pool.get(skey)
        if pool is None:
            if len(self._pool) >= 4:                    # ragged batch sizes: keep the latest ones' chains only
                torch.cuda.synchronize(self.device)
                for q in self._pool.values():
                    for inst in q:
                        self.lib.hint_chain_destroy(inst[0])
                    del q[:]
                self._pool.clear()
                self._wsbuf.clear()
            pool = self._pool[skey] = []
        inst = None
        while pool:
            cand = pool.pop()
            if cand[1] == key:
                inst = cand
                break
            torch.cuda.synchronize(self.device)         # an arena, packed buffer or permutation moved: stale
            self.lib.hint_chain_destroy(cand[0])
        if inst is None:
            ws = self._wsbuf.get(skey)
            handle, keep = self.build_chain(B, perms, G, ws=ws)
            self._wsbuf[skey] = keep[1]
            inst = (handle, key, keep)
        return inst, _Lease(pool, inst, self._drop_instance)

    def _drop_instance(self, inst):
        """a chain instance that does not go back to its (full) pool: its launches are through when its autograd node dies on
        the stream that ran them; destroy the handle (the tapes are plain tensors)"""
        torch.cuda.synchronize(self.device)
        self.lib.hint_chain_destroy(inst[0])


class _ChainFn(torch.autograd.Function):
    """autograd node of a whole HintFlow forward on the fused route: every block in one launch (hint_chain_forward), the
    backward in three (row-parallel part A, weight gradients, slab reduction: hint_chain_backward).  Parameter gradients are
    delivered as hint.py's 'direct' mode does, for the whole flow at once."""

    @staticmethod
    def forward(ctx, runner, x, c, anchor):
        B = x.shape[0]
        inst, lease = runner.acquire(B)
        z = torch.empty_like(x)
        J = torch.empty(B, dtype=torch.float32, device=x.device)
        runner.pack_all()           # (behind all host-side preparation: the re-pack and the forward launch go out back to back)
        with torch.cuda.device(runner.device):
            _lib.check(runner.lib.hint_chain_forward(inst[0], x.data_ptr(), c.data_ptr() if c is not None else None,
                                                     z.data_ptr(), J.data_ptr(), None, None,
                                                     torch.cuda.current_stream(runner.device).cuda_stream), "hint_chain_forward")
        ctx.runner, ctx.inst, ctx.lease = runner, inst, lease
        ctx.has_c = c is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, c) if c is not None else ctx.save_for_backward(x)
        return z, J

    @staticmethod
    def backward(ctx, gz, gJ):
        with _Region():
            runner = ctx.runner
            saved = ctx.saved_tensors
            x = saved[0]
            c = saved[1] if ctx.has_c else None
            gz = gz.contiguous() if gz is not None else torch.zeros_like(x)     # (part A of a chain wants g_z)
            gJ = gJ.contiguous() if gJ is not None else None
            gx = torch.empty_like(x)
            gc = torch.empty_like(c) if (c is not None and ctx.needs_input_grad[2]) else None
            chain = ctx.inst[0]
            ptr = lambda t: t.data_ptr() if t is not None else None

            def run(g_params, accumulate):
                # (the chain's blocks write their gradient slices of runner.G: g_params IS runner.G on this route)
                _mark_launch()
                with torch.cuda.device(runner.device):
                    _lib.check(runner.lib.hint_chain_backward(chain, x.data_ptr(), ptr(c), ptr(gz), ptr(gJ), gx.data_ptr(), ptr(gc),
                                                              1.0, 0.0, accumulate,
                                                              torch.cuda.current_stream(runner.device).cuda_stream),
                               "hint_chain_backward")

            G, views = runner.grad_views()
            params = runner.params
            grads = list(map(_GRAD, params))
            if all(map(operator.is_, grads, _NONES)):
                run(G, 0)
                for p, v in zip(params, views):
                    p.grad = v
            elif all(map(operator.is_, grads, views)):
                run(G, 1)
            else:
                # foreign gradients or a mix: the chain can only write G, so G's content (gradients some p.grad may alias)
                # is set aside, the fresh gradient copied out, and every parameter served as autograd would
                keep = G.clone()
                run(G, 0)
                fresh = G.clone()
                G.copy_(keep)
                vs = [v for e, (a, b) in zip(runner.engines, runner.slices) for v in e.split_flat(fresh[a:b])]
                for p, g, v in zip(params, grads, vs):
                    if g is None:
                        p.grad = v
                    else:
                        g.add_(v)
            return None, gx, gc, None


class HintFlow(nn.Module):
    """n coupling blocks chained by fixed orthogonal matrices (unconditional or with one
    condition input fed to every block, as conditional_recursive_cinn_4.py:58-70 does)."""

    def __init__(self, ndim_x: int, n_blocks: int, c_internal: Sequence[int], ndim_c: int = 0, clamp: float = 4.0,
                 max_splits: int = -1, min_split_size: int = 2, perm_seed: int = 1, perm_first: bool = False,
                 reshuffle: bool = False):
        super().__init__()
        self.ndim_x, self.ndim_c, self.n_blocks = ndim_x, ndim_c, n_blocks
        dims_c = [(ndim_c,)] if ndim_c > 0 else []
        self.perms = nn.ModuleList()
        self.blocks = nn.ModuleList()
        for i in range(n_blocks):
            has_perm = perm_first or i > 0        # power_hint_8.py:58 vs conditional_recursive_cinn_4.py:62
            self.perms.append(FixedOrthogonal([(ndim_x,)], seed=perm_seed + i) if has_perm else nn.Identity())
            self.blocks.append(HierarchicalAffineCouplingBlock([(ndim_x,)], dims_c=dims_c, c_internal=list(c_internal),
                                                               clamp=clamp, max_splits=max_splits,
                                                               min_split_size=min_split_size, reshuffle=reshuffle))
        self._jac = None
        # identical blocks on one GPU run as CHAINED launches (every block of the flow in one kernel per pass, the
        # permutations fused) instead of block by block; False: walk the modules one by one, as FrEIA's graph would
        self.fuse_chain = True
        self._runner: Optional[ChainRunner] = None

    # the runner (chain handles, device buffers: raw pointers) belongs to this object on its device
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_runner"] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        run, self._runner = self._runner, None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._runner = run
        return new

    def runner(self, device) -> ChainRunner:
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._runner is None or self._runner.device != device or \
                any(e is not blk.tree._engine for e, blk in zip(self._runner.engines, self.blocks)):
            self._runner = ChainRunner(self, device)
        return self._runner

    def _fused(self, x, cl, rev):
        """the whole flow in chained launches; None when this call has to walk the modules (not one CUDA device, blocks of
        different shapes, several condition tensors, gradients through rev=True, a partly frozen flow)"""
        if not self.fuse_chain or self.n_blocks == 0 or not isinstance(x, torch.Tensor) or not x.is_cuda \
                or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.ndim_x or x.shape[0] == 0 or len(cl) > 1:
            return None
        c = cl[0] if cl else None
        if (c is None) != (self.ndim_c == 0):
            return None
        if c is not None and (not c.is_cuda or c.dtype != torch.float32 or c.dim() != 2 or c.shape != (x.shape[0], self.ndim_c)):
            return None
        run = self.runner(x.device)
        if not run.chainable:
            return None
        grad = torch.is_grad_enabled()
        n_req = sum(map(_REQUIRES_GRAD, run.params)) if grad else 0
        needs = grad and (n_req > 0 or x.requires_grad or (c is not None and c.requires_grad))
        if needs and (rev or _hint._PARAM_GRADS != "direct" or n_req != len(run.params)):
            return None
        with _Region():
            x = x.contiguous()
            c = c.contiguous() if c is not None else None
            run.check_arenas()
            if needs:
                return _ChainFn.apply(run, x, c, run.anchor())
            B = x.shape[0]
            out = torch.empty_like(x)
            J = torch.empty(B, dtype=torch.float32, device=x.device)
            chain = run.chain_infer(B)
            run.pack_all()
            with torch.cuda.device(run.device):
                stream = torch.cuda.current_stream(run.device).cuda_stream
                cp = c.data_ptr() if c is not None else None
                if rev:
                    st = run.lib.hint_chain_inverse(chain, x.data_ptr(), cp, out.data_ptr(), J.data_ptr(), None, stream)
                else:
                    st = run.lib.hint_chain_forward(chain, x.data_ptr(), cp, out.data_ptr(), J.data_ptr(), None, None, stream)
            _lib.check(st, "hint_chain_inverse" if rev else "hint_chain_forward")
            return out, J

    def has_perm(self, i: int) -> bool:
        return isinstance(self.perms[i], FixedOrthogonal)

    @staticmethod
    def _unwrap(x):
        return x[0] if isinstance(x, (list, tuple)) else x

    def forward(self, x, c=None, rev=False):
        x = self._unwrap(x)
        cl = [] if c is None else ([self._unwrap(c)] if not isinstance(c, (list, tuple)) else list(c))
        fused = self._fused(x, cl, rev)
        if fused is not None:
            x, self._jac = fused
            return x
        jac = 0
        order = range(self.n_blocks) if not rev else reversed(range(self.n_blocks))
        for i in order:
            if not rev and self.has_perm(i):
                (x,) = self.perms[i]([x])
            (x,) = self.blocks[i]([x], c=cl, rev=rev)
            jac = jac + self.blocks[i].jacobian(None)
            if rev and self.has_perm(i):
                (x,) = self.perms[i]([x], rev=True)
        self._jac = jac
        return x

    def log_jacobian(self, x=None, c=None, rev=False, run_forward=True):
        if run_forward:
            if x is None:
                raise HintAmdError("log_jacobian(run_forward=True) needs x")
            self.forward(x, c=c, rev=rev)
        return self._jac
