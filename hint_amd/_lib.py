"""ctypes binding of libhint_amd.so (the C ABI declared in include/hint_amd.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the
caller gets an exception.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C hint_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HINT_AMD_LIB") or os.path.join(_HERE, "lib", "libhint_amd.so")
ABI_VERSION = 7


class HintAmdError(RuntimeError):
    pass


class NodeDesc(C.Structure):
    """mirror of `hint_node_desc` (include/hint_amd.h)"""
    _fields_ = [("off", C.c_int32), ("D", C.c_int32), ("k", C.c_int32), ("r", C.c_int32),
                ("h", C.c_int32), ("depth", C.c_int32), ("p_off", C.c_int64 * 12)]


_lib = None

_PROTOS = {
    "hint_abi_version": (C.c_int, []),
    "hint_last_error": (C.c_char_p, []),
    "hint_build_info": (C.c_char_p, []),
    "hint_debug_set_prefetch": (C.c_int, [C.c_int]),
    "hint_debug_reload_knobs": (C.c_int, []),
    "hint_debug_last_lds_bytes": (C.c_int32, [C.c_int32]),
    "hint_plan_create": (C.c_int, [C.POINTER(NodeDesc), C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                   C.POINTER(C.c_void_p)]),
    "hint_plan_check": (C.c_int, [C.POINTER(NodeDesc), C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                  C.POINTER(C.c_int64)]),
    "hint_plan_destroy": (None, [C.c_void_p]),
    "hint_plan_param_floats": (C.c_int64, [C.c_void_p]),
    "hint_plan_packed_floats": (C.c_int64, [C.c_void_p]),
    "hint_plan_tape_floats": (C.c_int64, [C.c_void_p, C.c_int32]),
    "hint_plan_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "hint_plan_lds_bytes": (C.c_int32, [C.c_void_p, C.c_int32]),
    "hint_plan_describe": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "hint_block_pack": (C.c_int, [C.c_void_p] * 4),
    "hint_pack_group_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.c_int32, C.POINTER(C.c_void_p)]),
    "hint_pack_group_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hint_pack_group_run_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hint_pack_group_destroy": (None, [C.c_void_p]),
    "hint_block_forward": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]),
    "hint_block_inverse": (C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_void_p]),
    "hint_block_backward": (C.c_int, [C.c_void_p] * 11 + [C.c_int32, C.c_void_p, C.c_size_t, C.c_int32,
                                                          C.c_void_p]),
    "hint_block_forward_ex": (C.c_int, [C.c_void_p] * 11 + [C.c_int32, C.c_void_p]),
    "hint_block_inverse_ex": (C.c_int, [C.c_void_p] * 9 + [C.c_int32, C.c_void_p]),
    "hint_block_backward_ex": (C.c_int, [C.c_void_p] * 11 + [C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                                             C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "hint_plan_inverse_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "hint_block_inverse_backward": (C.c_int, [C.c_void_p] * 9 + [C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32,
                                                                 C.c_void_p]),
    "hint_chain_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "hint_chain_set_block": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "hint_chain_commit": (C.c_int, [C.c_void_p]),
    "hint_chain_forward": (C.c_int, [C.c_void_p] * 8),
    "hint_chain_forward_noisy": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hint_chain_backward": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "hint_chain_destroy": (None, [C.c_void_p]),
    "hint_chain_backward_adam": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_float] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]
                                 + [C.c_float] * 6 + [C.c_void_p]),
    "hint_chain_backward_parts": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "hint_chain_wgrad_range": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hint_chain_inverse": (C.c_int, [C.c_void_p] * 7),
    "hint_chain_set_block_io": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hint_chain_wgrad_adam": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p]),
    "hint_block_forward_noisy": (C.c_int, [C.c_void_p] * 11 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "hint_block_backward_rows": (C.c_int, [C.c_void_p] * 11 + [C.c_size_t, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "hint_adam_step_dev": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_void_p] + [C.c_float] * 6 + [C.c_int32, C.c_void_p]),
    "hint_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32] + [C.c_float] * 7 + [C.c_int32,
                                                                                                 C.c_void_p]),
}


def exported_symbols():
    """names every entry point include/hint_amd.h declares (used by the symbol test)"""
    return sorted(_PROTOS)


def load():
    """Load the library once; raise HintAmdError loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HintAmdError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build() or `make -C hint_amd/csrc`). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. ROCm runtime missing
        raise HintAmdError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.hint_abi_version() != ABI_VERSION:
        raise HintAmdError(f"ABI version mismatch: library {lib.hint_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().hint_last_error()
        raise HintAmdError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
