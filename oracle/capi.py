"""ctypes access to the oracle's compiled pieces (TEST INFRASTRUCTURE — see palace_oracle.py).

  liboracle_c.so              C restatement (oracle_c.c), built by `make -C oracle`
  _ref/libpalace_qf_ref.so    the reference's own QFunction headers, built by `make -C oracle ref`
                              where /root/reference exists (this container); travels to the GPU
                              box as a prebuilt file.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
QF_HDIV, QF_HCURL, QF_HDIVMASS = 0, 1, 2


def build(ref: bool = True):
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/palace/fem/qfunctions"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _dp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_c.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
        _lib.oc_max_threads.restype = C.c_int
    return _lib


def build_geom_33(attr, qw, J):
    """attr [ne] float64, qw [Q], J [ne, 9, Q] -> geom [ne, 11, Q]."""
    ne, _, Q = J.shape
    geom = np.empty((ne, 11, Q))
    J = np.ascontiguousarray(J)
    lib().oc_build_geom_33(C.c_int(ne), C.c_int(Q), _dp(np.ascontiguousarray(attr, dtype=np.float64)),
                           _dp(np.ascontiguousarray(qw)), _dp(J), _dp(geom))
    return geom


def apply_add(off, ori, interp, deriv, geom, qf, ctx_blob, x, y, threads=None):
    """threads: OpenMP team size (default: 1 per 256 elements, capped at the core count)."""
    ne, P = off.shape
    if threads is None:
        threads = max(1, min(os.cpu_count() or 1, ne // 256))
    lib().oc_set_num_threads(C.c_int(threads))
    Q = geom.shape[2]
    off = np.ascontiguousarray(off, dtype=np.int32)
    ori8 = None if ori is None else np.ascontiguousarray(ori, dtype=np.uint8)
    lib().oc_apply_add(C.c_int(ne), C.c_int(P), C.c_int(Q), _dp(off), _dp(ori8),
                       _dp(np.ascontiguousarray(interp)), _dp(np.ascontiguousarray(deriv)),
                       _dp(np.ascontiguousarray(geom)), C.c_int(qf),
                       _dp(np.ascontiguousarray(ctx_blob)), _dp(x), _dp(y))
    return y


def qfunction(qf, ctx_blob, geom_e, u=None, cu=None):
    """One element: geom_e [11, Q]; u, cu [3, Q] -> (v, cv)."""
    Q = geom_e.shape[1]
    v = np.zeros((3, Q)) if qf in (QF_HCURL, QF_HDIVMASS) else None
    cv = np.zeros((3, Q)) if qf in (QF_HDIV, QF_HDIVMASS) else None
    lib().oc_qfunction(C.c_int(qf), _dp(np.ascontiguousarray(ctx_blob)), C.c_int(Q),
                       _dp(np.ascontiguousarray(geom_e)),
                       _dp(None if u is None else np.ascontiguousarray(u)),
                       _dp(None if cu is None else np.ascontiguousarray(cu)), _dp(v), _dp(cv))
    return v, cv


# ---- the real reference QFunctions (oracle/_ref) ---------------------------------------------

_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libpalace_qf_ref.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(_HERE, "_ref", "libpalace_qf_ref.so"))
    return _ref


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def ref_call(name, ctx_blob, Q, ins, outs):
    """Call reference QFunction `name(ctx, Q, in, out)`; ins / outs are lists of contiguous arrays."""
    fn = getattr(ref(), name)
    fn.restype = C.c_int
    ctxp = None if ctx_blob is None else np.ascontiguousarray(ctx_blob).ctypes.data_as(C.c_void_p)
    rc = fn(ctxp, C.c_int(Q), _ptr_array(ins), _ptr_array(outs))
    if rc != 0:
        raise RuntimeError(f"{name} returned {rc}")
