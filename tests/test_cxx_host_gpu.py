"""The C++ host layer (palace_amd/csrc/linalg.hpp) used directly from a C++ program, as Palace would:
build examples/cxx_host/solve.cpp with hipcc, run it, and compare with the same solve through the ctypes
mirror."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_cxx_host_solve(tmp_path):
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import ogrid_cylinder

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe, blob = str(tmp_path / "solve"), str(tmp_path / "problem.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "examples", "cxx_host", "dump_problem.py"), blob, "2"])
    libdir = os.path.join(ROOT, "palace_amd", "lib")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", "-w", "-I" + os.path.join(ROOT, "palace_amd", "csrc"),
                           os.path.join(ROOT, "examples", "cxx_host", "solve.cpp"), "-L" + libdir, "-lpalace_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.check_output([exe, blob], text=True)
    m = re.search(r"iterations (\d+)\s+converged (\d)\s+\|b - A x\| / \|b\| (\S+)\s+sum\(x\) (\S+)", out)
    assert m, out
    its, conv, res, sx = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    assert conv == 1 and res < 1e-8
    # the same solve through the ctypes mirror
    ctx = linalg.Context()
    mesh = ogrid_cylinder(2, 4)
    nd = NDHexSpace(mesh, 2)
    geom = ceed.GeomFactorData(mesh, 3)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    A = linalg.ParOperator(ctx, ceed.curlcurlmass_operator(geom, nd, mass, ceed.coefficient_context(3)), nd.ess_dofs(),
                           linalg.DIAG_ONE)
    K = linalg.cg(ctx, A, linalg.chebyshev(ctx, A, 4), rel_tol=1e-10, max_it=500)
    ones = torch.ones(nd.ndofs, dtype=torch.float64, device="cuda")
    b = torch.empty_like(ones)
    A.mult(ones, b)
    b[torch.from_numpy(nd.ess_dofs().astype(np.int64)).cuda()] = 0.0
    x = torch.zeros_like(b)
    K.mult(b, x)
    assert K.stats()["iterations"] == its
    assert abs(float(x.sum()) - sx) < 1e-9 * abs(sx)
