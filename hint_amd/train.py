"""Fast training step for a HintFlow on MI355X: no autograd graph, no per-parameter ops.

Reproduces one iteration of /root/reference/train_unconditional.py:114-144:
    x += 0.01*randn_like(x)                          (:121)
    z = model(x); J = model.log_jacobian(...)        (:124-125)
    loss = 0.5*sum(z^2,1).mean() - J.mean()          (:128-132)
    loss.backward(); clamp grads to +-5; Adam.step() (:137-144)
with the whole forward + backward chain as direct C-ABI kernel launches into model-wide flat
arenas (parameters / gradients / Adam moments), captured once in a hipGraph and replayed, then
ONE all-reduce of the gradient arena (hint_amd/dp.py) and ONE fused clamp+Adam launch.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import _lib, dp
from .flow import HintFlow
from .hint import HintAmdError


class _LossPair:
    """what step() returns: unpacks to the two loss terms, evaluated lazily so that no kernel
    is launched for them unless somebody looks"""

    def __init__(self, trainer):
        self._t = trainer

    def __iter__(self):
        return iter(self._t.last_losses())


class FlowTrainer:
    def __init__(self, flow: HintFlow, lr: float = 0.01 * 3e-2, betas=(0.9, 0.95), eps: float = 1e-4,
                 weight_decay: float = 1.86e-5, grad_clamp: float = 5.0, noise: float = 0.01,
                 use_graph: bool = True, group=None, use_chain: bool = True, seed: Optional[int] = None):
        self.lib = _lib.load()
        self.flow = flow
        self._lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.grad_clamp, self.noise = grad_clamp, noise
        self.use_graph = use_graph
        self.group = group
        self.step_count = 0
        dev = next(flow.parameters()).device
        if dev.type != "cuda":
            raise HintAmdError("FlowTrainer needs the flow on a GPU (no CPU path)")
        self.device = dev
        # model-wide arenas; every block's engine is bound to its slice
        self.engines = [blk.tree.engine(dev) for blk in flow.blocks]
        self.slices = []
        cursor = 0
        for e in self.engines:
            self.slices.append((cursor, cursor + e.total))
            cursor += e.total
        self.n_floats = cursor
        self.P = torch.zeros(cursor, dtype=torch.float32, device=dev)
        self.G = torch.zeros(cursor, dtype=torch.float32, device=dev)
        self.M = torch.zeros(cursor, dtype=torch.float32, device=dev)
        self.V = torch.zeros(cursor, dtype=torch.float32, device=dev)
        for e, (a, b) in zip(self.engines, self.slices):
            e.bind_external_arena(self.P[a:b])
            e.ensure_arena()
            e.pack()
        if dp.world_info(group)[1] > 1:
            # data-parallel replicas must start from the same weights (the reference idiom
            # p.data = init_scale*randn_like(p) draws per-process values): rank 0's win; M and V are zero
            src = torch.distributed.get_global_rank(group, 0) if group is not None else 0
            torch.distributed.broadcast(self.P, src=src, group=group)
            # ... and from the same buffers: the fixed permutations between the blocks and the node permutations of
            # reshuffle=True trees are drawn per process as well (hint.py:36-39 / power_hint_8.py:59-62), and the
            # kernels read them - replicas with different matrices would sum gradients of different functions
            for buf in flow.buffers():
                if buf.is_cuda and buf.numel() > 0:
                    torch.distributed.broadcast(buf, src=src, group=group)
            for e in self.engines:
                e._perm_key = None         # (composed permutations are rebuilt from the received matrices)
                e.pack()
        self.loss_acc = torch.zeros(64, 2, dtype=torch.float32, device=dev)   # per-slot partial loss sums
        self._loss_single = self.loss_acc      # (step_many points loss_acc at its last iteration's sums)
        # chain handles, the one-launch re-pack and the inference chains (sample()) are the flow's ChainRunner's
        self._runner = flow.runner(dev)
        self._runner.perms()                    # (makes the permutation matrices contiguous: the kernels read raw pointers)
        self._graph = None
        self._static = None
        # identical blocks (the configs stack copies of one block) run as ONE forward launch and
        # TWO backward launches for the whole flow (hint_chain_*)
        self._chainable = use_chain and self._runner.chainable
        # one chain (handle, tapes, workspaces) per batch size: a captured graph keeps raw pointers into
        # its chain's buffers, so a chain is never destroyed while a graph that used it is alive
        self._chains = {}
        # state of the in-kernel noise generator (hint_chain_forward_noisy): {seed, step}; every rank
        # of a data-parallel job draws its own stream
        if seed is None:            # (not from torch's global generator: building a trainer must not shift the caller's random stream)
            seed = int.from_bytes(os.urandom(8), "little") >> 2
        rank = dp.world_info(group)[0] if hasattr(dp, "world_info") else 0
        self.rng_state = torch.tensor([(seed + 0x9E3779B97F4A7C15 * rank) & (2 ** 63 - 1), 0], dtype=torch.int64,
                                      device=dev)
        # Adam's hyper-parameters and per-step factors in device memory: with one process (no
        # all-reduce between backward and optimizer) the fused clamp+Adam launch is captured in the
        # step's graph and reads them there (hint_adam_step_dev); rng_state[1] is the step count
        self.opt_state = torch.tensor([lr, betas[0], betas[1], 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32,
                                      device=dev)
        self._adam_in_graph = False
        self._allreduce_in_graph = False
        # data-parallel jobs all-reduce the gradient arena in ONE bucket behind the backward pass (north_star: "a
        # single RCCL all-reduce of gradients per step").  HINT_DP_BUCKETS=2: two buckets - the second half of the blocks
        # (whose part B then runs as a launch of its own, first) goes out on a side stream while the weight gradients of
        # the first half are still being computed; never measured with more than one rank, hence not the default
        nb = len(self.engines)
        self._split = nb // 2 if (self._chainable and nb >= 2 and os.environ.get("HINT_DP_BUCKETS", "1") == "2") else 0
        self._side = None
        self._warming = False       # _capture's warm-up passes issue no collectives (a re-capture on one rank must not hang the others)

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):          # learning-rate schedules (train_unconditional.py:191-199) need no re-capture
        self._lr = float(value)
        self.opt_state[0] = self._lr

    def __del__(self):
        try:
            self._graph = self._graph_many = None
            for handle, _, _ in getattr(self, "_chains", {}).values():
                self.lib.hint_chain_destroy(handle)
            self._chains = {}
        except Exception:
            pass

    def _chain_for(self, B: int):
        """the chain handle for batch size B: tapes and backward workspaces of all blocks are
        allocated once and the pointer table uploaded (ChainRunner.build_chain); rebuilt when B or any buffer moved"""
        run = self._runner
        perms = run.perms()
        key = run.chain_key(B, perms, self.G)
        have = self._chains.get(B)
        if have is not None and have[1] == key:
            return have[0]
        if have is not None:
            # an arena, packed buffer or permutation moved: every captured graph may point at the old
            # addresses - drop them all (they are re-captured on their next use), then the stale chain
            self._graph = self._graph_many = None
            self._static = None
            torch.cuda.synchronize(self.device)        # (launches that read its device table may still be queued)
            self.lib.hint_chain_destroy(have[0])
            del self._chains[B]
        elif len(self._chains) >= 8:
            # many batch sizes (ragged data): forget the chains no live graph was captured on
            keep = {self._static["x"].shape[0]} if self._graph is not None and self._static is not None else set()
            st = getattr(self, "_static_many", None)
            if getattr(self, "_graph_many", None) is not None and st is not None:
                keep.add(st["x"].shape[1])
            torch.cuda.synchronize(self.device)
            for b in [b for b in self._chains if b not in keep]:
                self.lib.hint_chain_destroy(self._chains[b][0])
                del self._chains[b]
        handle, bufs = run.build_chain(B, perms, self.G)
        self._chains[B] = (handle, key, bufs)
        return handle

    def _chain_infer(self, B: int):
        """the chain handle of the sampling / evaluation direction for batch size B (no tape, no workspace)"""
        return self._runner.chain_infer(B)

    def _dp_overlap(self) -> bool:
        """the bucket all-reduces are issued from inside the backward pass (side stream) when the backend is RCCL
        (stream-ordered, capturable); with gloo (CPU tests, two ranks on one GPU) they follow the backward pass"""
        return torch.distributed.is_available() and torch.distributed.is_initialized() and self.P.is_cuda and \
            torch.distributed.get_backend(self.group) == "nccl" and os.environ.get("HINT_DP_OVERLAP", "1") != "0"

    # ---- the un-captured step body ----------------------------------------------------
    def _adam_dev(self, scale: float = 1.0):
        """the fused clamp+Adam launch with its step factors read from opt_state (capturable: nothing in
        it depends on the host's step count); scale = 1/world turns the all-reduced sum into the mean"""
        with torch.cuda.device(self.device):
            st = self.lib.hint_adam_step_dev(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                             self.n_floats, self.opt_state.data_ptr(), self.betas[0], self.betas[1],
                                             self.eps, self.wd, scale, self.grad_clamp, 1,
                                             torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(st, "hint_adam_step_dev")

    def _fwd_bwd(self, x: torch.Tensor, c: Optional[torch.Tensor], with_adam: bool = False):
        """pack -> forward chain -> backward chain, every piece a direct C-ABI launch: the fixed
        permutations, the running log-det, the two loss sums and the loss gradient are folded
        into the block kernels (hint_block_*_ex)."""
        flow, B = self.flow, x.shape[0]
        self._reduced = False
        # weights of the previous optimizer step -> MFMA order; the same launch clears the loss sums
        # and advances the noise counter
        self._pack_all(step_prologue=True)
        if self._chainable and B > 0:
            chain = self._chain_for(B)
            z = torch.empty_like(x)
            J = torch.empty(B, dtype=torch.float32, device=x.device)
            gx = torch.empty_like(x)
            cp = c.data_ptr() if c is not None else None
            noisy = self.noise > 0
            xn = torch.empty_like(x) if noisy else x          # the perturbed input, for the backward pass
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device).cuda_stream
                _lib.check(self.lib.hint_chain_forward_noisy(
                    chain, x.data_ptr(), cp, z.data_ptr(), J.data_ptr(), None, self.loss_acc.data_ptr(),
                    float(self.noise), self.rng_state.data_ptr() if noisy else None, xn.data_ptr() if noisy else None,
                    stream), "hint_chain_forward_noisy")
                # dL/dz = z / B and dL/dJ = -1/B: applied inside the kernel
                dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
                if dist_on and self._split > 0:
                    # part A, then part B in two launches: blocks [h, n) first - their slice of G is final behind it and its
                    # all-reduce starts on a side stream (overlap: RCCL only; gloo blocks the host) - then blocks [0, h)
                    h, n = self._split, len(self.engines)
                    cut = self.slices[h][0]
                    _lib.check(self.lib.hint_chain_backward_parts(chain, xn.data_ptr(), cp, z.data_ptr(), None, gx.data_ptr(),
                                                                  None, 1.0 / B, -1.0 / B, 1, 1, stream), "hint_chain_backward_parts")
                    _lib.check(self.lib.hint_chain_wgrad_range(chain, xn.data_ptr(), cp, 1, h, n, stream), "hint_chain_wgrad_range")
                    # (inside a graph capture only when the collectives are captured as well)
                    overlap = self._dp_overlap() and not self._warming and \
                        (self._allreduce_in_graph or not torch.cuda.is_current_stream_capturing())
                    if overlap:
                        if self._side is None:
                            self._side = torch.cuda.Stream(device=self.device)
                        cur = torch.cuda.current_stream(self.device)
                        self._side.wait_stream(cur)
                        with torch.cuda.stream(self._side):
                            torch.distributed.all_reduce(self.G[cut:], op=torch.distributed.ReduceOp.SUM, group=self.group)
                    _lib.check(self.lib.hint_chain_wgrad_range(chain, xn.data_ptr(), cp, 1, 0, h, stream), "hint_chain_wgrad_range")
                    if overlap:
                        torch.distributed.all_reduce(self.G[:cut], op=torch.distributed.ReduceOp.SUM, group=self.group)
                        torch.cuda.current_stream(self.device).wait_stream(self._side)
                        self._reduced = True      # step() must not reduce again
                elif with_adam and not self._allreduce_in_graph and os.environ.get("HINT_FUSE_ADAM", "1") != "0":
                    # one process: the clamp + Adam step rides in the weight gradients' final reduction (the gradient arena
                    # stays zero, as the separate optimizer launch leaves it)
                    _lib.check(self.lib.hint_chain_backward_adam(
                        chain, xn.data_ptr(), cp, z.data_ptr(), None, gx.data_ptr(), None, 1.0 / B, -1.0 / B,
                        self.P.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.n_floats, self.opt_state.data_ptr(),
                        self.betas[0], self.betas[1], self.eps, self.wd, 1.0, self.grad_clamp, stream), "hint_chain_backward_adam")
                    with_adam = False
                else:
                    _lib.check(self.lib.hint_chain_backward(chain, xn.data_ptr(), cp, z.data_ptr(), None, gx.data_ptr(),
                                                            None, 1.0 / B, -1.0 / B, 1, stream), "hint_chain_backward")
            if with_adam:
                if self._allreduce_in_graph:      # data-parallel job: the RCCL all-reduces are captured as well
                    if not getattr(self, "_reduced", False):
                        torch.distributed.all_reduce(self.G, op=torch.distributed.ReduceOp.SUM, group=self.group)
                    self._adam_dev(1.0 / dp.world_info(self.group)[1])
                else:
                    self._adam_dev()
            return B
        if self.noise > 0:
            x = x.add(torch.randn_like(x), alpha=self.noise)
        inputs, tapes = [], []
        h, J = x, None
        n = len(self.engines)
        for i, eng in enumerate(self.engines):
            perm = eng.compose_perm(flow.perms[i].W if flow.has_perm(i) else None)
            inputs.append(h if perm is None else None)
            h, J, tape = eng.forward_chain(h, c, perm, J, self.loss_acc if i == n - 1 else None, with_tape=True)
            tapes.append(tape)
        z = h
        g = z                                  # dL/dz = z / B : the scale is applied inside the kernel
        for i in reversed(range(n)):
            a, b = self.slices[i]
            perm = self.engines[i].compose_perm(flow.perms[i].W if flow.has_perm(i) else None)
            g = self.engines[i].backward_chain(inputs[i], tapes[i], c, g, (1.0 / B) if i == n - 1 else 1.0,
                                               -1.0 / B, perm, self.G[a:b], accumulate=True)
        return B

    def _pack_all(self, step_prologue: bool = False):
        """one launch re-packs every block (ChainRunner.pack_all).  step_prologue: the launch also zeroes the loss sums,
        advances the noise counter and writes Adam's factors of the step."""
        if step_prologue:
            self._runner.pack_all(self.loss_acc, self.rng_state, self.opt_state)
        else:
            self._runner.pack_all()

    def _check_arenas(self):
        """parameters rebound from outside (p.data = ..., load_state_dict into new storage)
        are pulled back into the arena and re-packed"""
        for e in self.engines:
            before = e.arena
            e.ensure_arena()
            if e.arena is not before or e._regathered:
                e._regathered = False
                e.pack()

    def repack(self):
        """call after changing parameters in place from outside the trainer"""
        for e in self.engines:
            e.ensure_arena()
            e.pack()

    def _optimizer(self, grad_scale: float):
        self.step_count += 1
        with torch.cuda.device(self.device):
            st = self.lib.hint_adam_step(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                         self.n_floats, self.step_count, self.lr, self.betas[0], self.betas[1],
                                         self.eps, self.wd, grad_scale, self.grad_clamp, 1,
                                         torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(st, "hint_adam_step")

    def step(self, x: torch.Tensor, c: Optional[torch.Tensor] = None):
        """one training iteration on this rank's shard; returns device scalars (l0, l1) =
        ('-log p(z)', '-log |det J|') of the LOCAL shard (train_unconditional.py:162)"""
        self.loss_acc = self._loss_single
        if not self.use_graph:
            self._check_arenas()
            self._fwd_bwd(x, c)
        else:
            if self._graph is not None and any(e.params[0].data_ptr() != e._ptrs[0] for e in self.engines):
                self._graph = None             # parameters were rebound from outside (p.data = ..., .to()): re-capture
            if self._graph is None or self._static["x"].shape != x.shape:
                self._capture(x, c)
            # (a batch that already sits in the graph's input buffers - input_buffers() - is not copied again)
            if x.data_ptr() != self._static["x"].data_ptr():
                self._static["x"].copy_(x)
            if c is not None and c.data_ptr() != self._static["c"].data_ptr():
                self._static["c"].copy_(c)
            self._graph.replay()
        self._last_B = x.shape[0]
        for e in self.engines:                 # the step's kernels changed the weights in place: packed copies are stale
            e._pack_key = None
        if self.use_graph and self._adam_in_graph:
            self.step_count += 1               # the optimizer ran inside the graph
        else:
            if getattr(self, "_reduced", False) and not self.use_graph:
                scale = 1.0 / dp.world_info(self.group)[1]     # (the eager backward pass issued the bucket all-reduces itself)
            elif self._split > 0 and torch.distributed.is_available() and torch.distributed.is_initialized():
                cut = self.slices[self._split][0]              # the same two buckets, one after the other
                dp.allreduce_sum_(self.G[cut:], self.group)
                scale = dp.allreduce_sum_(self.G[:cut], self.group)
            else:
                scale = dp.allreduce_sum_(self.G, self.group)
            self._optimizer(scale)
        return _LossPair(self)

    # ---- several iterations per graph replay ----------------------------------------------------
    def _many_ok(self) -> bool:
        return self.use_graph and self._chainable and dp.world_info(self.group)[1] == 1 and \
            not (torch.distributed.is_available() and torch.distributed.is_initialized())

    def _capture_many(self, xs: torch.Tensor, cs: Optional[torch.Tensor]):
        """K = xs.shape[0] consecutive iterations (re-pack, forward, backward, clamp+Adam each) in ONE
        hipGraph: the ~8 us between the end of one graph replay and the start of the next are paid once
        per K steps.  One process only (the gradient all-reduce of a data-parallel job is not captured)."""
        K = xs.shape[0]
        self._check_arenas()
        sx = xs.clone()
        sc = cs.clone() if cs is not None else None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # warm-up on a side stream (allocator, plan LDS attrs); no optimizer
            for _ in range(2):
                self._fwd_bwd(sx[0], sc[0] if sc is not None else None)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.G.zero_()
        scratch = torch.zeros(4, 4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):   # load the optimizer kernel outside the capture
            self.lib.hint_adam_step(scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(),
                                    scratch[3].data_ptr(), 4, 1, 0.0, 0.9, 0.95, 1e-4, 0.0, 1.0, 0.0, 0,
                                    torch.cuda.current_stream(self.device).cuda_stream)
        self.rng_state[1] = self.step_count
        self._loss_all = torch.zeros(K, 64, 2, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(K):
                self.loss_acc = self._loss_all[k]             # every iteration its own loss sums
                self._fwd_bwd(sx[k], sc[k] if sc is not None else None, with_adam=True)
        self._graph_many, self._static_many = g, dict(x=sx, c=sc)

    def step_many(self, xs: torch.Tensor, cs: Optional[torch.Tensor] = None):
        """K = xs.shape[0] training iterations on the batches xs[k] ([K, B, d]; cs [K, B, dc]); with
        use_graph and one process they are ONE graph replay.  Per-iteration losses: step_losses()."""
        K, B = xs.shape[0], xs.shape[1]
        if not self._many_ok():
            out = []
            for k in range(K):
                self.step(xs[k], cs[k] if cs is not None else None)
                out.append(torch.stack(self.last_losses()))
            self._many_losses = torch.stack(out)
            return
        st = getattr(self, "_static_many", None)
        if getattr(self, "_graph_many", None) is None or st["x"].shape != xs.shape:
            self._capture_many(xs, cs)
            st = self._static_many
        if xs.data_ptr() != st["x"].data_ptr():
            st["x"].copy_(xs)
        if cs is not None and cs.data_ptr() != st["c"].data_ptr():
            st["c"].copy_(cs)
        self._graph_many.replay()
        for e in self.engines:
            e._pack_key = None
        self.loss_acc = self._loss_all[K - 1]
        self.step_count += K
        self._last_B = B
        self._many_losses = None

    def step_losses(self) -> torch.Tensor:
        """[K, 2] device tensor: (-log p(z), -log|det J|) of every iteration of the last step_many()"""
        if getattr(self, "_many_losses", None) is not None:
            return self._many_losses
        s = self._loss_all.sum(dim=1)
        return torch.stack([s[:, 0] / self._last_B, -s[:, 1] / self._last_B], dim=1)

    def input_buffers_many(self, xs: torch.Tensor, cs: Optional[torch.Tensor] = None):
        """step_many()'s own input tensors ([K, B, d], [K, B, dc]) holding a copy of the arguments (see input_buffers)"""
        if not self._many_ok():
            return xs, cs
        st = getattr(self, "_static_many", None)
        if getattr(self, "_graph_many", None) is None or st["x"].shape != xs.shape:
            self._capture_many(xs, cs)
            st = self._static_many
        st["x"].copy_(xs)
        if cs is not None:
            st["c"].copy_(cs)
        return st["x"], st["c"]

    def input_buffers(self, x: torch.Tensor, c: Optional[torch.Tensor] = None):
        """the captured step's own input tensors (x, c) for this batch shape, holding a copy of the
        arguments: a data pipeline that writes its batches straight into them (index_select(..., out=),
        copy_ from pinned memory) and passes them to step() saves the device-to-device copy step()
        otherwise makes per iteration.  Without use_graph the arguments are returned as they are."""
        if not self.use_graph:
            return x, c
        if self._graph is None or self._static["x"].shape != x.shape:
            self._capture(x, c)
        self._static["x"].copy_(x)
        if c is not None:
            self._static["c"].copy_(c)
        return self._static["x"], self._static["c"]

    def timed_step(self, x: torch.Tensor, c: Optional[torch.Tensor] = None):
        """one un-captured training step with HIP events between the launches (on the stream they
        are issued to); returns {launch: microseconds}.  For bench.py's roofline: the durations are
        those of the kernels inside a real step (weights just re-packed, caches as they are), not
        of a hot back-to-back loop."""
        if not self._chainable:
            raise HintAmdError("timed_step needs the chained launches (identical blocks)")
        self._check_arenas()
        B = x.shape[0]
        chain = self._chain_for(B)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        z = torch.empty_like(x); J = torch.empty(B, dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x); xn = torch.empty_like(x)
        cp = c.data_ptr() if c is not None else None
        noisy = self.noise > 0
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            ev[0].record()
            self._pack_all(step_prologue=True)
            ev[1].record()
            _lib.check(self.lib.hint_chain_forward_noisy(
                chain, x.data_ptr(), cp, z.data_ptr(), J.data_ptr(), None, self.loss_acc.data_ptr(), float(self.noise),
                self.rng_state.data_ptr() if noisy else None, xn.data_ptr() if noisy else None, stream), "forward")
            ev[2].record()
            xin = xn if noisy else x
            for k, parts in ((3, 1), (4, 2)):         # part A (row-parallel), then part B (weight gradients)
                _lib.check(self.lib.hint_chain_backward_parts(chain, xin.data_ptr(), cp, z.data_ptr(), None, gx.data_ptr(),
                                                              None, 1.0 / B, -1.0 / B, 1, parts, stream), "backward")
                ev[k].record()
            self._last_B = B
            scale = dp.allreduce_sum_(self.G, self.group)
            self._optimizer(scale)
            ev[5].record()
        torch.cuda.synchronize(self.device)
        fwd, bwd = self.kernel_names(B)
        names = ["hint_pack_many_kernel", fwd, bwd, "hint_wgrad_kernel+hint_wreduce_kernel", "allreduce+hint_adam_kernel"]
        return {n: ev[i].elapsed_time(ev[i + 1]) * 1e3 for i, n in enumerate(names)}

    def allreduce_plan(self) -> str:
        """what the step does with the gradient arena between backward and optimizer (for bench.py's config line)"""
        world = dp.world_info(self.group)[1]
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        if not dist_on:
            return "none (one process: clamp + Adam ride in the weight gradients' final reduction)"
        backend = torch.distributed.get_backend(self.group)
        where = "captured in the step's hipGraph" if self._allreduce_in_graph else "issued from the host after the graph replay"
        if self._split > 0:
            return (f"two buckets over {world} ranks ({backend}): blocks [{self._split}, {len(self.engines)}) first, beside the "
                    f"rest of part B (HINT_DP_BUCKETS=2); {where}")
        return f"one all-reduce (sum) of the flat fp32 gradient arena ({self.n_floats} floats) over {world} ranks ({backend}); {where}"

    def kernel_names(self, B: int):
        """(forward, backward part A) kernel of a batch of B rows as rocprof prints them: the wave-local kernels for narrow
        trees (template arguments: direction / row tiles per workgroup), the general ones otherwise"""
        import ctypes as C
        info = (C.c_int32 * 8)()
        _lib.check(self.lib.hint_plan_describe(self.engines[0].plan, B, info), "hint_plan_describe")
        if info[0]:     # (third template argument: a chained launch - the trainer's)
            return f"hint_wl_apply_kernel<false, {info[1]}, true>", f"hint_wl_bwd_kernel<{info[1]}, true>"
        if info[7] and os.environ.get("HINT_NO_BWD_FLY", "0") in ("", "0"):
            bwd = "hint_bwd_kernel_fly"        # plans with lean general groups (hint_abi.cpp run_backward)
        else:
            bwd = "hint_bwd_kernel_n3" if info[5] <= 3 and not info[6] else "hint_bwd_kernel"
        return f"hint_apply_kernel<false, {'true' if info[7] else 'false'}>", bwd

    def _capture(self, x, c):
        self._check_arenas()
        sx = x.clone()
        sc = c.clone() if c is not None else None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self._warming = True
        try:
            with torch.cuda.stream(side):      # warm-up on a side stream (allocator, plan LDS attrs); no collectives
                for _ in range(2):
                    self._fwd_bwd(sx, sc)
        finally:
            self._warming = False
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.G.zero_()                         # the warm-up runs accumulated into the gradient arena
        torch.cuda.synchronize(self.device)
        # one process, identical blocks: the optimizer launch goes into the graph as well (no host
        # gap between the weight-gradient kernel and Adam).  Its kernel has been loaded by a launch
        # outside the capture; the device step counter is aligned with the host's.
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        # RCCL collectives can be captured (backend "nccl"): the whole data-parallel iteration - backward,
        # gradient all-reduce, clamp+Adam - is then one graph replay per rank (HINT_GRAPH_ALLREDUCE=0: off)
        self._allreduce_in_graph = self._chainable and dist_on and self.P.is_cuda and \
            torch.distributed.get_backend(self.group) == "nccl" and os.environ.get("HINT_GRAPH_ALLREDUCE", "1") != "0"
        self._adam_in_graph = self._chainable and ((dp.world_info(self.group)[1] == 1 and not dist_on) or self._allreduce_in_graph)
        if self._allreduce_in_graph:           # communicator set up and used once outside the capture
            warm = torch.zeros(8, dtype=torch.float32, device=self.device)
            torch.distributed.all_reduce(warm, group=self.group)
            torch.cuda.synchronize(self.device)
        if self._adam_in_graph:
            scratch = torch.zeros(4, 4, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                self.lib.hint_adam_step(scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(),
                                        scratch[3].data_ptr(), 4, 1, 0.0, 0.9, 0.95, 1e-4, 0.0, 1.0, 0.0, 0,
                                        torch.cuda.current_stream(self.device).cuda_stream)
        self.rng_state[1] = self.step_count
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        try:
            # (thread_local: the process group's watchdog thread may touch the HIP runtime meanwhile)
            with torch.cuda.graph(g, capture_error_mode="thread_local" if self._allreduce_in_graph else "global"):
                self._fwd_bwd(sx, sc, with_adam=self._adam_in_graph)
        except Exception:
            if not self._allreduce_in_graph:
                raise
            # the collective could not be captured here: keep it (and the optimizer) outside the graph
            self._allreduce_in_graph = self._adam_in_graph = False
            torch.cuda.synchronize(self.device)
            self.G.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fwd_bwd(sx, sc, with_adam=False)
        self._graph = g
        self._static = dict(x=sx, c=sc)

    def last_losses(self):
        """(-log p(z), -log|det J|) of the most recent step's local shard as device scalars
        (train_unconditional.py:162 labels).  The sums live in a buffer the next step overwrites:
        read them before stepping again."""
        s = self.loss_acc.sum(dim=0)
        return s[0] / self._last_B, -s[1] / self._last_B

    @torch.no_grad()
    def sample(self, z: torch.Tensor, c: Optional[torch.Tensor] = None):
        """x = f^-1(z) and the log-determinant of that map per row (train_unconditional.py:152-153,
        rev=True through the whole graph; hint.py:83 sign) - every block of the flow in ONE launch
        (hint_chain_inverse), on the weights as they are now."""
        if not self._chainable:
            x = self.flow(z, c=c, rev=True)
            return x, self.flow.log_jacobian(rev=True, run_forward=False)
        z = z.contiguous().float()
        B = z.shape[0]
        x = torch.empty_like(z)
        J = torch.empty(B, dtype=torch.float32, device=z.device)
        if B == 0:
            return x, J
        self._check_arenas()
        self._pack_all()
        chain = self._chain_infer(B)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.hint_chain_inverse(chain, z.data_ptr(), c.data_ptr() if c is not None else None, x.data_ptr(),
                                                   J.data_ptr(), None, torch.cuda.current_stream(self.device).cuda_stream),
                       "hint_chain_inverse")
        return x, J

    @torch.no_grad()
    def nll(self, x: torch.Tensor, c: Optional[torch.Tensor] = None) -> float:
        """mean negative log-likelihood in nats incl. the Gaussian constant
        (run_uci_experiments.py:71-72)"""
        z = self.flow(x, c=c)
        J = self.flow.log_jacobian(run_forward=False)
        return float(0.5 * torch.sum(z * z, dim=1).mean() - J.mean()) + 0.5 * self.flow.ndim_x * math.log(2 * math.pi)
