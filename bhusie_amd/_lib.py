"""Loader for the in-tree libbhray.so.  There is no fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

# BHRAY_LIB selects another in-tree build of the SAME sources (kernel tuning variants); never a fallback.
LIB_PATH = os.environ.get("BHRAY_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbhray.so")


class LibraryMissing(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C bhusie_amd/csrc`).  bhusie_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        from .layouts import declare
        declare(L)
        _lib = L
    return _lib
