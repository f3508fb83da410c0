"""Shared pieces of bench.py's legs: constants, the oracle's inputs for a problem (checker), small helpers."""
import json  # noqa: F401
import os
import sys
import time  # noqa: F401

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


_ORACLE_CACHE = {}


def oracle_hex_data(prob, order):
    """The C oracle's inputs for the finest space of a SlabProblem (geometry data, restriction, dense tables), built once per
    problem: several legs check their device results against it at the full size."""
    key = (id(prob), order)
    if key not in _ORACLE_CACHE:
        from oracle import capi
        from oracle import palace_oracle as po
        from tests import util

        capi.build(ref=False)
        nd = prob.spaces[-1]
        off, ori = nd.native_restriction()
        interp, curl = po.nd_hex_dense_tables(order, order + 1, nd.dof_map_native())
        _ORACLE_CACHE.clear()  # (one problem at a time: the geometry data of the 10M-dof mesh is 0.7 GB)
        _ORACLE_CACHE[key] = dict(geom=util.oracle_geom(prob.mesh, order + 1), off=off, ori=ori, interp=interp, curl=curl)
    return _ORACLE_CACHE[key]


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, 64)  # the element loop stops scaling beyond a socket's worth of threads
