"""The Palace-side shim of INTEGRATION.md section 1 -- the class a maintainer would put in place of palace::ceed::Operator
(fem/libceed/operator.hpp:32-65) -- is compiled from the document's own text against a stand-in for mfem::Vector /
mfem::Operator (tests/cpu/fake_mfem.hpp) and run: on the CPU the calls that need no device, on the GPU every apply form
(Mult, AddMult, MultTranspose, AssembleDiagonal, and the SetDofMultiplicity-scaled forms of operator.cpp:181-240) against the
C oracle."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim_exe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    lib = os.path.join(ROOT, "palace_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libpalace_amd.so")):
        import __graft_entry__ as ge
        ge.build()
    d = tmp_path_factory.mktemp("shim")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```cpp\n(// palace/fem/amd/operator\.hpp.*?)```", text, re.S)
    assert block, "INTEGRATION.md section 1 lost its shim"
    (d / "integration_shim.hpp").write_text(block.group(1))
    exe = str(d / "integration_shim_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-unused-result", "-I" + str(d),
                           "-I" + os.path.join(ROOT, "tests", "cpu"), "-I" + os.path.join(ROOT, "include"),
                           "-x", "hip", os.path.join(ROOT, "tests", "cpu", "integration_shim_check.cpp"),
                           "-L" + lib, "-lpalace_amd", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_documented_shim_compiles_and_runs_without_a_device(shim_exe):
    out = subprocess.run([shim_exe], capture_output=True, text=True)
    assert out.returncode == 0 and "shim ok (no device)" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_documented_shim_applies_match_the_oracle(shim_exe, tmp_path):
    from palace_amd import ceed
    from palace_amd.fem.basis1d import gauss_legendre
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import ogrid_cylinder
    from tests import util

    mesh = ogrid_cylinder(2, 3)
    p, q1d = 2, 3
    nd = NDHexSpace(mesh, p)
    _, bm = util.make_ctx("aniso", nattr=int(mesh.attr.max()))
    _, bc = util.make_ctx("scalar", nattr=int(mesh.attr.max()))
    blob = np.concatenate([bm, bc])
    off, ori = nd.native_restriction()
    qx, qw = gauss_legendre(q1d)
    B, G = ceed._q2_1d(qx)
    t = ceed.Tables1D(p, q1d)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, nd.ndofs)
    d = rng.uniform(0.25, 1.0, nd.ndofs)
    head = np.array([mesh.ne, 2, q1d, mesh.x.shape[0], nd.P, nd.ndofs, ceed.FE_HCURL, p, ceed.QF_HDIVMASS_33,
                     ceed.EVAL_CURL | ceed.EVAL_INTERP, ceed.EVAL_CURL | ceed.EVAL_INTERP], dtype=np.int32)
    arrays = [head, mesh.elem_nodes.astype(np.int32), mesh.x.astype(np.float64), mesh.attr.astype(np.int32), B, G, qw,
              off.astype(np.int32), ori.astype(np.uint8), t.Bc, t.Gc, t.Bo, np.asarray(nd.dof_map_native(), dtype=np.int32),
              np.frombuffer(np.ascontiguousarray(blob).tobytes(), dtype=np.uint8), x, d]
    path = str(tmp_path / "problem.bin")
    with open(path, "wb") as f:
        f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
        for a in arrays:
            a = np.ascontiguousarray(a)
            f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
            f.write(a.tobytes())
    out = subprocess.run([shim_exe, path], capture_output=True, text=True)
    assert out.returncode == 0 and "shim ok (device)" in out.stdout, out.stdout + out.stderr
    y, y2, yt, diag, ym, yma, ymt = np.fromfile(path + ".out", dtype=np.float64).reshape(7, nd.ndofs)
    og = util.oracle_geom(mesh, q1d)
    ax = util.oracle_apply_c(nd, og, "hdivmass", blob, x, q1d)
    scale = np.linalg.norm(ax)

    def rel(a, b):
        return np.linalg.norm(a - b) / scale

    assert rel(y, ax) < 1e-12 and rel(y2, 2 * ax) < 1e-12 and rel(yt, ax) < 1e-12  # (symmetric coefficients: A^T = A)
    assert rel(ym, d * ax) < 1e-12 and rel(yma, x + d * ax) < 1e-12
    assert rel(ymt, util.oracle_apply_c(nd, og, "hdivmass", blob, d * x, q1d)) < 1e-12
    # the diagonal: e_i^T A e_i through the oracle on a sample of dofs
    for i in rng.choice(nd.ndofs, 12, replace=False):
        e = np.zeros(nd.ndofs)
        e[i] = 1.0
        assert abs(util.oracle_apply_c(nd, og, "hdivmass", blob, e, q1d)[i] - diag[i]) < 1e-12 * abs(diag).max()
