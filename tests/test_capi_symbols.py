"""The C-ABI library loads and exports every symbol include/palace_amd*.h declares (no compute)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    from palace_amd import lib

    L = lib.load()
    declared = _declared()
    assert len(declared) >= 15
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.pa_version()


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from palace_amd import lib

    L = lib.load()
    assert L.pa_device_count() == 0
    g = ctypes.c_void_p()

    class Dummy(ctypes.Structure):
        _fields_ = lib.MeshDesc._fields_

    d = lib.MeshDesc(1, 2, 3, 27, None, None, None, None, None, None)
    rc = L.pa_geom_create(ctypes.byref(d), None, ctypes.byref(g))
    assert rc != 0 and b"no HIP device" in L.pa_last_error()
