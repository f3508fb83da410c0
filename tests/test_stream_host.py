"""Host-side encodings of the streaming H(curl) hex kernel (palace_amd/csrc/pa_stream_host.hpp), checked on CPU:
the packed index words / slot bytes / flags and the run form of the transpose map against a plain C++ model of the
kernels' decode paths (tests/cpu/stream_host_check.cpp), and the run statistics on the reference's cylinder mesh."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpu", "stream_host_check.cpp")
INC = os.path.join(ROOT, "palace_amd", "csrc")


def test_encodings_against_cpu_model(tmp_path):
    exe = str(tmp_path / "stream_host_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + INC, SRC, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_run_statistics_on_cylinder(tmp_path, cylinder_mesh):
    """Order-3 Nedelec dofs on the 80-hex27 cylinder mesh: faces collapse to runs of 12, edges to runs of 3."""
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import refine_uniform

    so = str(tmp_path / "libstream_host_check.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + INC, SRC, "-o", so])
    lib = ctypes.CDLL(so)
    mesh = refine_uniform(cylinder_mesh)
    nd = NDHexSpace(mesh, 3)
    off, ori = nd.native_restriction()
    dm = np.asarray(nd.dof_map_native(), dtype=np.int64)
    P = off.shape[1] if off.ndim == 2 else nd.P
    off = np.asarray(off).reshape(mesh.ne, P)
    ori = np.asarray(ori).reshape(mesh.ne, P).astype(bool)
    nat = np.where(dm >= 0, dm, -1 - dm)
    neg = (dm < 0)[None, :] ^ ori[:, nat]
    d = off[:, nat]
    lidx = np.ascontiguousarray(np.where(neg, -1 - d, d).astype(np.int32))
    ns, ncp = ctypes.c_int(), ctypes.c_int()
    nruns = lib.stream_host_run_stats(mesh.ne, P, nd.ndofs, lidx.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ns),
                                      ctypes.byref(ncp))
    assert nruns > 0
    # interior face: 12 dofs / run, edge: 3 dofs / run  =>  well above 4 dofs per run on average
    assert ns.value / nruns > 4.0, (ns.value, nruns)
    print(f"{mesh.ne} elements, {ns.value} shared dofs in {nruns} runs ({ns.value / nruns:.2f} dofs/run), "
          f"{ncp.value} run copies")

