"""The hand-counted waits behind hint_sub.hpp's untracked prefetch loads (inline-asm global loads the compiler's wait-count
bookkeeping does not see) are only correct while no instruction touches their destination registers before the matching
`s_waitcnt vmcnt(N)` and while N vector-memory operations really are younger on every path.  `make gate` compiles the kernels
that contain them to assembly (no GPU needed) and tools/check_untracked_loads.py proves both properties path by path; the
checker itself is tested on three tiny hand-written listings."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_untracked_loads_are_waited_for_on_every_path():
    import shutil
    import pytest
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if shutil.which(hipcc) is None:
        pytest.skip(f"{hipcc} not found: the gate compiles the kernels to assembly")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "hint_amd", "csrc"), "gate"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "untracked loads checked" in r.stdout
    checked = [int(l.split(":")[1].split()[0]) for l in r.stdout.splitlines() if "untracked loads checked" in l]
    assert sum(checked) >= 12, r.stdout        # (the subtree phases of the forward, inverse and both backward instances)


def _check(listing, tmp_path):
    import check_untracked_loads as chk
    f = tmp_path / "k.s"
    f.write_text(textwrap.dedent(listing))
    errs = []
    for name, ins in chk.parse(str(f)):
        errs += chk.check_function(str(f), name, ins)[1]
    return errs


def test_checker_accepts_a_covered_load_and_flags_an_uncovered_one(tmp_path):
    good = """
    _Zkernel:
    \t;;#ASMSTART
    \tglobal_load_dwordx4 v[10:13], v[2:3], off
    \t;;#ASMEND
    \tglobal_store_dword v[4:5], v6, off
    \tglobal_store_dword v[4:5], v7, off
    \t;;#ASMSTART
    \ts_waitcnt vmcnt(2)
    \t;;#ASMEND
    \tv_mov_b32_e32 v20, v10
    \ts_endpgm
    """
    assert _check(good, tmp_path) == []
    early_use = good.replace("\tglobal_store_dword v[4:5], v7, off\n", "\tv_mov_b32_e32 v20, v11\n")
    assert any("touches" in e for e in _check(early_use, tmp_path))
    # the second store sits under a branch: on the path around it only one operation is younger - vmcnt(2) proves nothing
    skipped = good.replace("\tglobal_store_dword v[4:5], v7, off\n", "\ts_cbranch_execz .LBB0_1\n\tglobal_store_dword v[4:5], v7, off\n.LBB0_1:\n")
    assert any("touches" in e for e in _check(skipped, tmp_path))


def test_checker_flags_a_load_in_flight_at_the_end(tmp_path):
    bad = """
    _Zkernel:
    \t;;#ASMSTART
    \tglobal_load_dwordx4 v[10:13], v[2:3], off
    \t;;#ASMEND
    \ts_endpgm
    """
    assert any("in flight" in e for e in _check(bad, tmp_path))
