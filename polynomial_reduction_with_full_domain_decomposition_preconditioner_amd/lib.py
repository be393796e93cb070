"""ctypes loader for the C-ABI libraries (plumbing only).

`libfdd_hip.so` (include/fdd_hip.h) holds the gfx950 kernels; `libfdd_host.so`
(include/fdd_host.h) holds the C++ host classes that mirror the reference's
CSR_Matrix / Math / Domain / Subdomain on top of it.  Signatures are taken
from the headers themselves, so every declared entry point is bound and a
missing symbol fails at load time.

There is NO fallback: if a library is missing or a call returns non-zero this
module raises.  The CPU oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
INCLUDE_DIR = os.path.join(REPO_ROOT, "include")


class FddError(RuntimeError):
    pass


_SCALARS = {
    "int": ctypes.c_int,
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "size_t": ctypes.c_size_t,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "long": ctypes.c_long,
}


def _ctype_of(decl: str):
    """C parameter declaration -> ctypes type.  Every pointer is a c_void_p;
    `T *const name[N]` / `const T *const name[N]` is a host array of pointers."""
    decl = decl.strip()
    if decl == "void":
        return None
    if "[" in decl or "*" in decl:
        return ctypes.c_void_p
    words = [w for w in decl.replace("const", " ").split() if w]
    # last word is the parameter name
    type_words = words[:-1] if len(words) > 1 else words
    tname = " ".join(type_words)
    if tname in _SCALARS:
        return _SCALARS[tname]
    if tname.endswith("_fn"):  # function-pointer typedefs
        return ctypes.c_void_p
    raise FddError(f"cannot map C parameter '{decl}'")


_DECL_RE = re.compile(r"^\s*(int|size_t|const char \*)\s*(\w+)\s*\(([^;{]*)\)\s*;", re.M | re.S)


def parse_header(path: str) -> Dict[str, Tuple[object, List[object]]]:
    """name -> (restype, argtypes) for every function declared in a C-ABI header."""
    with open(path) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out: Dict[str, Tuple[object, List[object]]] = {}
    for m in _DECL_RE.finditer(text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char *": ctypes.c_char_p}[ret]
        args: List[object] = []
        params = " ".join(params.split())
        if params and params != "void":
            for p in params.split(","):
                t = _ctype_of(p)
                if t is not None:
                    args.append(t)
        out[name] = (restype, args)
    return out


class _Lib:
    def __init__(self, so_path: str, header: str, err_fn: str):
        if not os.path.exists(so_path):
            raise FddError(
                f"{so_path} is missing: the HIP extension is not built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                f"There is no CPU fallback."
            )
        self.path = so_path
        self.cdll = ctypes.CDLL(so_path, mode=ctypes.RTLD_GLOBAL)
        self.decls = parse_header(header)
        self._err_fn = err_fn
        for name, (restype, argtypes) in self.decls.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as exc:
                raise FddError(f"{so_path} does not export {name} declared in {header}") from exc
            fn.restype = restype
            fn.argtypes = argtypes

    def raw(self, name: str):
        return getattr(self.cdll, name)

    def call(self, name: str, *args):
        """Call an int-returning entry; raise on non-zero."""
        fn = getattr(self.cdll, name)
        rc = fn(*args)
        if self.decls[name][0] is ctypes.c_int and rc != 0:
            msg = getattr(self.cdll, self._err_fn)()
            raise FddError(f"{name} failed with code {rc}: {msg.decode() if msg else ''}")
        return rc


_hip = None
_host = None


def hip() -> _Lib:
    """libfdd_hip.so.  torch is imported first so that the library resolves the
    HIP runtime torch already loaded (one runtime per process)."""
    global _hip
    if _hip is None:
        import torch  # noqa: F401  (loads libamdhip64 with RTLD_GLOBAL)

        _hip = _Lib(os.path.join(PKG_DIR, "libfdd_hip.so"), os.path.join(INCLUDE_DIR, "fdd_hip.h"), "fdd_last_error")
    return _hip


def host() -> _Lib:
    """libfdd_host.so (C++ host classes over libfdd_hip.so)."""
    global _host
    if _host is None:
        hip()
        _host = _Lib(os.path.join(PKG_DIR, "libfdd_host.so"), os.path.join(INCLUDE_DIR, "fdd_host.h"), "fddh_last_error")
    return _host


def ptr(x) -> ctypes.c_void_p:
    """torch tensor / numpy array / int / None -> c_void_p."""
    if x is None:
        return ctypes.c_void_p(0)
    if isinstance(x, ctypes.c_void_p):
        return x
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return ctypes.c_void_p(x.ctypes.data)
    raise TypeError(f"cannot take a pointer of {type(x)}")


def ptr_array(items) -> ctypes.Array:
    """HOST array of device pointers (G[6], GDu[3], D_hat_ptr[levels])."""
    arr = (ctypes.c_void_p * len(items))()
    for k, it in enumerate(items):
        arr[k] = ptr(it).value
    return arr


def current_stream() -> ctypes.c_void_p:
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
