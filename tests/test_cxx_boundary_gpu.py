"""A bilinear form with domain and boundary integrators through the C++ front end (palace_amd/csrc/fem.hpp: BilinearForm::
AddDomainIntegrator / AddBoundaryIntegrator on dense-table spaces, integrators choosing their QFunction by (space_dim, dim)
like fem/integ/*.cpp): examples/cxx_host/boundary_form.cpp assembles K(mu^-1) + M(eps) on the tetrahedra and the surface
mass sigma and surface curl-curl lambda on the boundary triangles (models/spaceoperator.cpp:270-303) as ONE operator; its
action and diagonal are compared with the sum of the oracle's operators."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "cxx_host"))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("cxx_bdr")
    out = str(d / "boundary_form")
    libdir = os.path.join(ROOT, "palace_amd", "lib")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", "-w", "-I" + os.path.join(ROOT, "palace_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "cxx_host", "boundary_form.cpp"),
                           "-L" + libdir, "-lpalace_amd", "-Wl,-rpath," + libdir, "-o", out])
    return out, d


@pytest.mark.parametrize("curved", [0, 1])
@pytest.mark.parametrize("p", [1, 2])
def test_cxx_domain_and_boundary_form(exe, p, curved):
    import dump_boundary_problem as dp

    binary, d = exe
    blob, out = str(d / f"problem{p}{curved}.bin"), str(d / f"y{p}{curved}.bin")
    dp.main(blob, p, 2, curved)
    r = subprocess.run([binary, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout and "symmetric 1" in r.stdout, r.stdout + r.stderr
    P = dp.problem(p, 2, curved)
    m, nd, blk = P["mesh"], P["nd"], P["blk"]
    J = m.jacobians(P["pts"])
    og = po.build_geom_factor_33(m.attr.astype(np.float64), P["wts"], np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 9))
    Jb = blk.jacobians(P["bpts"])
    ogb = po.build_geom_factor_32(blk.attr.astype(np.float64), P["bwts"], np.transpose(Jb, (0, 1, 3, 2)).reshape(blk.ne, -1, 6))

    def ctx(mats, dim=3):
        return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.asarray(a) for a in mats], dim=dim)

    kw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    vol = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, P["nint"], P["ncurl"], og,
                                po.QF_HDIVMASS, ctx(P["eps"]), ctx(P["muinv"]), **kw)
    smass = po.CeedOperatorOracle(nd.ndofs, blk.offsets, blk.orients, P["bint"], P["bcurl"], ogb, po.QF_HCURL_32, ctx(P["sigma"]))
    scurl = po.CeedOperatorOracle(nd.ndofs, blk.offsets, blk.orients, P["bint"], P["bcurl"], ogb, po.QF_L2_1,
                                  ctx([np.array([v]) for v in P["lam"]], dim=1), qw=P["bwts"])
    n = nd.ndofs
    got = np.fromfile(out, dtype=np.float64).reshape(2, n)
    y_ref = sum(o.apply_add(P["x"], np.zeros(n)) for o in (vol, smass, scurl))
    d_ref = sum(o.diagonal() for o in (vol, smass, scurl))
    assert np.abs(got[0] - y_ref).max() < 1e-12 * np.abs(y_ref).max()
    assert np.abs(got[1] - d_ref).max() < 1e-12 * np.abs(d_ref).max()
    # the boundary terms matter: without them the result differs at the 1e-2 level
    y_vol = vol.apply_add(P["x"], np.zeros(n))
    assert np.abs(y_ref - y_vol).max() > 1e-3 * np.abs(y_ref).max()
