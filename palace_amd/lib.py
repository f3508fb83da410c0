"""ctypes binding of libpalace_amd.so (the C ABI in include/palace_amd.h).

There is no fallback: if the HIP library is missing or no GPU is visible, compute entry points
raise.  `load()` only needs the shared object (used by the CPU-side symbol tests)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PALACE_AMD_LIB", os.path.join(_HERE, "lib", "libpalace_amd.so"))

_lib = None


class PalaceAmdError(RuntimeError):
    pass


class RestrictionDesc(C.Structure):
    _fields_ = [("num_elem", C.c_int32), ("elem_size", C.c_int32), ("lsize", C.c_int32),
                ("offsets", C.c_void_p), ("orients", C.c_void_p), ("curl_orients", C.c_void_p)]


class DenseBasisDesc(C.Structure):
    _fields_ = [("fe_type", C.c_int32), ("num_dofs", C.c_int32), ("num_qpts", C.c_int32),
                ("interp", C.c_void_p), ("deriv", C.c_void_p)]


class MeshDenseDesc(C.Structure):
    _fields_ = [("num_elem", C.c_int32), ("nodes_per_elem", C.c_int32), ("num_qpts", C.c_int32),
                ("num_nodes", C.c_int32), ("node_offsets", C.c_void_p), ("nodes", C.c_void_p),
                ("attr", C.c_void_p), ("mesh_grad", C.c_void_p), ("qweight", C.c_void_p), ("dim", C.c_int32),
                ("space_dim", C.c_int32)]


class BasisDesc(C.Structure):
    _fields_ = [("fe_type", C.c_int32), ("order", C.c_int32), ("q1d", C.c_int32),
                ("Bc", C.c_void_p), ("Gc", C.c_void_p), ("Bo", C.c_void_p),
                ("dof_map", C.c_void_p), ("interp", C.c_void_p), ("deriv", C.c_void_p)]


class MeshDesc(C.Structure):
    _fields_ = [("num_elem", C.c_int32), ("mesh_order", C.c_int32), ("q1d", C.c_int32),
                ("num_nodes", C.c_int32), ("node_offsets", C.c_void_p), ("nodes", C.c_void_p),
                ("attr", C.c_void_p), ("mesh_B", C.c_void_p), ("mesh_G", C.c_void_p),
                ("qweight1d", C.c_void_p)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PalaceAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950); there is no CPU fallback")
    # torch owns the device context and ships its own HIP runtime: load it first so that this library
    # binds to the runtime already in the process instead of bringing a second one
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    lib.pa_last_error.restype = C.c_char_p
    lib.pa_version.restype = C.c_char_p
    lib.pa_op_algorithmic_bytes.restype = C.c_double
    lib.pa_op_algorithmic_bytes.argtypes = [C.c_void_p]
    for name in ("pa_geom_destroy", "pa_op_destroy", "pa_error_op_destroy"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [C.c_void_p]
    _lib = lib
    return lib


def last_error():
    """The message of this thread's last failed call (pa_last_error)."""
    return load().pa_last_error().decode()


def check(rc):
    if rc != 0:
        raise PalaceAmdError(last_error())
