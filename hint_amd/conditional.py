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
