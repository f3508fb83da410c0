"""The C++ flux error estimators (palace_amd/csrc/errorestimator.hpp: FluxProjector, GradFluxErrorEstimator,
CurlFluxErrorEstimator, TimeDependentFluxErrorEstimator, ErrorIndicator -- linalg/errorestimator.cpp:111-541,
fem/errorindicator.cpp:11-47) used from a C++ program on dense-table tetrahedral spaces: build examples/cxx_host/estimate.cpp
with hipcc, run it on a dumped problem and compare the element indicators with the same procedure carried out with the
oracle's operators and dense solves."""
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


def _build(tmp_path_factory, name):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("cxx_" + name)
    out = str(d / name)
    libdir = os.path.join(ROOT, "palace_amd", "lib")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", "-w", "-I" + os.path.join(ROOT, "palace_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "cxx_host", name + ".cpp"),
                           "-L" + libdir, "-lpalace_amd", "-Wl,-rpath," + libdir, "-o", out])
    return out, d


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return _build(tmp_path_factory, "estimate")


@pytest.fixture(scope="module")
def exe2d(tmp_path_factory):
    return _build(tmp_path_factory, "estimate2d")


def _sym_fun(M, f):
    w, V = np.linalg.eigh(M)
    return (V * f(w)) @ V.T


def _oracle_indicators(P, Et):
    m, nd, sp = P["mesh"], P["nd"], P["rt"]
    J = m.jacobians(P["pts"])
    og = po.build_geom_factor_33(m.attr.astype(np.float64), P["wts"], np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 9))
    cid = po.CoeffCtx()
    kw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, P["nint"], P["ncurl"], og,
                                po.QF_HCURL, cid, **kw)
    rto = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, P["rint"], P["rint"], og, po.QF_HDIV, cid)
    Mn = np.stack([ndo.apply_add(e, np.zeros(ndo.lsize)) for e in np.eye(ndo.lsize)], axis=1)
    Mr = np.stack([rto.apply_add(e, np.zeros(rto.lsize)) for e in np.eye(rto.lsize)], axis=1)

    def ctx(mats):
        return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=list(mats))

    eps, mui = P["eps"], P["muinv"]
    D = np.linalg.solve(Mr, po.MixedSpaceOracle(ndo, rto, og, po.QF_HCURLHDIV, ctx(eps)).apply_add(P["E"], np.zeros(rto.lsize)))
    eg = po.MixedSpaceOracle(ndo, rto, og, po.QF_HCURLHDIV_ERROR, ctx([_sym_fun(e, np.sqrt) for e in eps]),
                             ctx([_sym_fun(e, lambda w: w ** -0.5) for e in eps])).error_add(P["E"], D, np.zeros(m.ne))
    H = np.linalg.solve(Mn, po.MixedSpaceOracle(rto, ndo, og, po.QF_HDIVHCURL, ctx(mui)).apply_add(P["B"], np.zeros(ndo.lsize)))
    ec = po.MixedSpaceOracle(rto, ndo, og, po.QF_HDIVHCURL_ERROR, ctx([_sym_fun(e, np.sqrt) for e in mui]),
                             ctx([_sym_fun(e, lambda w: w ** -0.5) for e in mui])).error_add(P["B"], H, np.zeros(m.ne))
    s1, s2 = np.sqrt((eg + ec) * 0.5 / Et), np.sqrt(eg + ec)
    return np.sqrt(eg * 0.5 / Et), np.sqrt(ec * 0.5 / Et), np.sqrt((s1 ** 2 + s2 ** 2) / 2)


@pytest.mark.parametrize("curved", [0, 1])
def test_cxx_flux_error_estimators(exe, curved):
    import dump_estimator_problem as dp

    binary, d = exe
    blob, out = str(d / f"problem{curved}.bin"), str(d / f"ind{curved}.bin")
    dp.main(blob, 2, 2, curved)
    r = subprocess.run([binary, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    P = dp.problem(2, 2, curved)
    ne = P["mesh"].ne
    got = np.fromfile(out, dtype=np.float64).reshape(3, ne)
    ref = _oracle_indicators(P, 0.37)
    for g, e, name in zip(got, ref, ("grad", "curl", "time-dependent")):
        assert e.min() > 0
        assert np.abs(g - e).max() < 1e-8 * e.max(), (name, np.abs(g - e).max(), e.max())
    # MixedVectorGradientIntegrator through the C++ front end (BilinearForm(h1, nd | rt)) vs the oracle
    m, nd, sp, h1 = P["mesh"], P["nd"], P["rt"], P["h1"]
    J = m.jacobians(P["pts"])
    og = po.build_geom_factor_33(m.attr.astype(np.float64), P["wts"], np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 9))
    kw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, P["nint"], P["ncurl"], og,
                                po.QF_HCURL, None, **kw)
    rto = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, P["rint"], P["rint"], og, po.QF_HDIV, None)
    h1o = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, P["hint"], P["hgrad"], og, po.QF_HCURL, None, vector_fe=False)
    c_eps = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=list(P["eps"]))
    gg = np.fromfile(out + ".grad", dtype=np.float64)
    for got_g, (to, qfo) in zip((gg[: nd.ndofs], gg[nd.ndofs :]), ((ndo, po.QF_HCURL), (rto, po.QF_HCURLHDIV))):
        ref_g = po.MixedSpaceOracle(h1o, to, og, qfo, c_eps, first_tab=h1o.deriv).apply_add(P["phi"], np.zeros(to.lsize))
        assert np.abs(got_g - ref_g).max() < 1e-12 * np.abs(ref_g).max(), qfo
    norms = [float(l.split()[2]) for l in r.stdout.splitlines() if "norm" in l]
    for n, e in zip(norms, ref):
        assert abs(n - np.linalg.norm(e)) < 1e-8 * np.linalg.norm(e)
    # GradientIntegrator through the C++ front end (VectorFiniteElementSpace, byNODES then byVDIM) vs the oracle
    gv = np.fromfile(out + ".vgrad", dtype=np.float64).reshape(2, 3 * h1.ndofs)
    ref_v = po.MixedSpaceOracle(h1o, h1o, og, po.QF_HCURLH1D, c_eps, first_tab=h1o.deriv).gradient_add(
        P["phi"], np.zeros(3 * h1.ndofs), h1.ndofs)
    assert np.abs(gv[0] - ref_v).max() < 1e-12 * np.abs(ref_v).max()
    assert np.abs(gv[1] - ref_v.reshape(3, h1.ndofs).T.ravel()).max() < 1e-12 * np.abs(ref_v).max()
    # MixedVectorWeakDivergenceIntegrator -(eps E, grad v) and the vector H1 mass (f_apply_h1_3) through the C++ front end
    wv = np.fromfile(out + ".wdiv", dtype=np.float64)
    c_neg = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=list(P["eps"]), a=-1.0)
    ref_w = po.MixedSpaceOracle(ndo, h1o, og, po.QF_HCURL, c_neg, second_tab=h1o.deriv).apply_add(P["E"], np.zeros(h1.ndofs))
    assert np.abs(wv[: h1.ndofs] - ref_w).max() < 1e-12 * np.abs(ref_w).max()
    uq = np.einsum("qj,ej->eq", h1o.interp[0], P["phi"][h1.offsets])  # values of phi at the points
    vq = po.apply_h1_vec(c_eps, og, np.stack([uq, 2 * uq, 3 * uq], axis=1))
    ref_m = np.zeros(3 * h1.ndofs)
    for c in range(3):
        np.add.at(ref_m, (c * h1.ndofs + h1.offsets).ravel(), np.einsum("qj,eq->ej", h1o.interp[0], vq[:, c, :]).ravel())
    assert np.abs(wv[h1.ndofs :] - ref_m).max() < 1e-12 * np.abs(ref_m).max()
    # DivDivMassIntegrator through the C++ front end (BilinearForm(rt), f_apply_l2mass_33) vs the oracle, and vs the sum of
    # DivDivIntegrator + VectorFEMassIntegrator assembled next to it
    yy = np.fromfile(out + ".divdivmass", dtype=np.float64).reshape(2, sp.ndofs)
    c_mass = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=list(P["muinv"]))
    c_div = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    ref_p = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, P["rint"], P["rdiv"], og, po.QF_L2MASS, c_mass, c_div,
                                  qw=P["wts"], deriv_comps=1).apply_add(P["B"], np.zeros(sp.ndofs))
    assert np.abs(yy[0] - ref_p).max() < 1e-12 * np.abs(ref_p).max()
    assert np.abs(yy[1] - ref_p).max() < 1e-12 * np.abs(ref_p).max()


def test_cxx_plane_flux_error_estimators(exe2d):
    """The plane branch (errorestimator.cpp:343-349, :446-472) through the C++ classes: GradFluxErrorEstimator with 2 x 2
    tensors on (ND, RT) and CurlFluxErrorEstimator with the scalar curl in a discontinuous space, its H1 recovery and 1 x 1
    tensors, on the triangles of the reference's cavity2d mesh; against the oracle operators with sparse direct solves."""
    import scipy.sparse.linalg as spla

    import dump_estimator_problem_2d as dp

    binary, d = exe2d
    blob, out = str(d / "problem2d.bin"), str(d / "ind2d.bin")
    dp.main(blob, 1)
    r = subprocess.run([binary, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    P = dp.problem(1)
    m, nd, h1 = P["mesh"], P["nd"], P["h1"]
    J = m.jacobians(P["pts"])
    og = po.build_geom_factor_22(m.attr.astype(np.float64), P["wts"], np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 4))
    c2, c1 = po.CoeffCtx(dim=2), po.CoeffCtx(dim=1)
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, P["nint"], P["ncurl"], og, po.QF_HCURL_22, c2, qw=P["wts"])
    rto = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, P["rint"], P["ncurl"], og, po.QF_HDIV_22, c2, qw=P["wts"])
    h1o = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, P["hint"], P["hgrad"], og, po.QF_H1MASS, c1, vector_fe=False)
    l2o = po.CeedOperatorOracle(P["l2_off"].size, P["l2_off"], None, P["hint"], P["hgrad"], og, po.QF_H1MASS, c1, vector_fe=False)

    def ctx(mats, dim):
        return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=list(mats), dim=dim)

    eps, mui = P["eps"], P["muinv"]
    D = spla.spsolve(rto.assemble_sparse().tocsc(),
                     po.MixedSpaceOracle(ndo, rto, og, po.QF_HCURLHDIV_22, ctx(eps, 2)).apply_add(P["E"], np.zeros(rto.lsize)))
    eg = po.MixedSpaceOracle(ndo, rto, og, po.QF_HCURLHDIV_ERROR_22, ctx([_sym_fun(e, np.sqrt) for e in eps], 2),
                             ctx([_sym_fun(e, lambda w: w ** -0.5) for e in eps], 2)).error_add(P["E"], D, np.zeros(m.ne))
    H = spla.spsolve(h1o.assemble_sparse().tocsc(),
                     po.MixedSpaceOracle(l2o, h1o, og, po.QF_H1MASS, ctx(mui, 1)).apply_add(P["B"], np.zeros(h1o.lsize)))
    ec = po.MixedSpaceOracle(l2o, h1o, og, po.QF_L2H1_ERROR, ctx([np.sqrt(e) for e in mui], 1),
                             ctx([e ** -0.5 for e in mui], 1)).error_add(P["B"], H, np.zeros(m.ne))
    got = np.fromfile(out, dtype=np.float64).reshape(2, m.ne)
    for g, e, name in zip(got, (np.sqrt(eg * 0.5 / 0.37), np.sqrt(ec * 0.5 / 0.37)), ("grad", "curl")):
        assert e.min() > 0
        assert np.abs(g - e).max() < 1e-8 * e.max(), (name, np.abs(g - e).max(), e.max())
