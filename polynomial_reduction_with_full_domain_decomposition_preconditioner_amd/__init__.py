"""MI355X-native preconditioned-CG hot path for the SEM Poisson solve.

Product layers (no CPU fallback anywhere):
  csrc/   gfx950 HIP kernels + C-ABI  -> libfdd_hip.so   (include/fdd_hip.h)
  host/   C++ host classes mirroring the reference's CSR_Matrix / Math /
          Domain / Subdomain and the poisson driver -> libfdd_host.so, poisson
  *.py    ctypes plumbing used by tests/, bench.py and __graft_entry__.py
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
