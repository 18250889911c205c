"""Randomised sweeps as part of the GPU suite (fixed seeds; the tools take a case count and a seed):
block-level parity against the float32 oracle and flow-level consistency of the chained launches with the
autograd path, over random lanes / widths / conditions / batch sizes / split rules / permutations."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_parity.py", 60, 0), ("fuzz_flow.py", 30, 0)])
def test_randomised_sweep(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "worst:" in r.stdout
