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

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .hint import HierarchicalAffineCouplingBlock, HintAmdError


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

    def has_perm(self, i: int) -> bool:
        return isinstance(self.perms[i], FixedOrthogonal)

    @staticmethod
    def _unwrap(x):
        return x[0] if isinstance(x, (list, tuple)) else x

    def forward(self, x, c=None, rev=False):
        x = self._unwrap(x)
        cl = [] if c is None else ([self._unwrap(c)] if not isinstance(c, (list, tuple)) else list(c))
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
