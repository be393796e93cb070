"""Thin call helper over libfdd_hip.so for tests and bench: torch tensors in,
device pointers out.  Not a compute path of its own."""
from __future__ import annotations

import ctypes

from . import lib


def k(name: str, *args):
    """Call C-ABI entry `name`.  Tensors become device pointers, lists/tuples of
    tensors become host arrays of device pointers, and the trailing `stream`
    argument is filled with torch's current stream when omitted."""
    L = lib.hip()
    restype, argtypes = L.decls[name]
    conv = []
    keep = []  # host pointer tables must outlive the call
    for a in args:
        if isinstance(a, (list, tuple)):
            arr = lib.ptr_array(a)
            keep.append(arr)
            conv.append(ctypes.cast(arr, ctypes.c_void_p))
        elif hasattr(a, "data_ptr") or a is None:
            conv.append(lib.ptr(a))
        else:
            conv.append(a)
    if len(conv) == len(argtypes) - 1:
        conv.append(lib.current_stream())
    if len(conv) != len(argtypes):
        raise TypeError(f"{name} takes {len(argtypes)} arguments, got {len(conv)}")
    return L.call(name, *conv)


def reduce_workspace(device="cuda"):
    import torch

    n = lib.hip().raw("fdd_reduce_workspace_doubles")()
    return torch.empty(n, dtype=torch.float64, device=device)
