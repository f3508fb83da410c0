"""Config 3 on the reference's own mesh: examples/cpw/mesh/cpw_lumped_0.msh (14 628 tet4, silicon substrate + air, the
trace as an internal PEC surface), order-3 Nedelec tetrahedra with the curl-oriented restriction, the driven-type complex
system A = K - k0^2 eps_r (1 - i tan d) M at 16 GHz solved on the device by FGMRES with the Hiptmair p-multigrid
(levels p = 1, 2, 3) built on the shifted real matrix K + k0^2 eps_r M -- the reference's configuration for driven
problems (models/spaceoperator.cpp:316-331, linalg/ksp.cpp).  The device solution is checked with the ORACLE's operators
(dense tables, native curl-oriented restriction, reference QFunction arithmetic): ||A_oracle x - b|| <= 1e-6 ||b||, and
the device and oracle applies of both operator parts agree to 1e-12."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_cpw_complex_fgmres_against_oracle_operator():
    from oracle import palace_oracle as po
    from palace_amd import ceed, linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cpw_mesh.npz"))
    mesh = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
    p = 3
    ctx = linalg.Context()
    prob = TetProblem(ctx, mesh, p)  # spaces p = 1..3, geometry data with the order-2p rule of the fine level
    # PEC: far field (4) and the metal trace (13, an internal surface); the port faces stay natural
    bt = np.sort(np.asarray(d["bdr_tris"], dtype=np.int64), axis=1)
    fkey = {tuple(f): i for i, f in enumerate(map(tuple, mesh.face_verts))}
    fmask = np.zeros(mesh.face_verts.shape[0], dtype=bool)
    fmask[[fkey[tuple(f)] for f in bt[np.isin(d["bdr_attr"], (4, 13))]]] = True
    ess = [s.ess_dofs(fmask) for s in prob.spaces]
    nd, n = prob.spaces[-1], prob.spaces[-1].ndofs
    k0 = 2 * np.pi * 16.0e9 * 1.0e-6 / 299792458.0  # 16 GHz, mesh in micrometres (cpw_lumped_uniform.json: L0 = 1e-6)
    eps, tand = np.array([1.0, 11.7]), np.array([0.0, 0.05])  # attributes 1 air, 2 si (lossy for the test)

    def coef(vals):
        return ceed.coefficient_context(3, attr_mat=[0, 1], mat_coeff=[np.array([vals[0]]), np.array([vals[1]])])

    ident = ceed.coefficient_context(3)
    blocks = [prob.nd_block(s) for s in prob.spaces]

    def nd_op(block, qf, blob, ops):
        return ceed.Operator(block.lsize, block.lsize).add_dense_integrator(prob.geom, block, qf, blob, ops).finalize()

    Kr = nd_op(blocks[-1], ceed.QF_HDIVMASS_33, np.concatenate([coef(-k0 ** 2 * eps), ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP)
    Ki = nd_op(blocks[-1], ceed.QF_HCURL_33, coef(k0 ** 2 * eps * tand), ceed.EVAL_INTERP)
    A = linalg.ComplexParOperator(ctx, Kr, Ki, ess[-1], linalg.DIAG_ONE)

    # the oracle's view of the same two operators
    pts, wts = prob.pts, prob.wts
    interp, curl = nd.elem.tables(pts)
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    okw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)

    def ocoef(vals):
        return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([vals[0]]), np.array([vals[1]])])

    oKr = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HDIVMASS, ocoef(-k0 ** 2 * eps), po.CoeffCtx(), **okw)
    oKi = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HCURL, ocoef(k0 ** 2 * eps * tand), **okw)
    e = ess[-1]

    def oA(v):  # ComplexParOperator::Mult on the oracle (rap.cpp:483-519)
        t = v.copy()
        t[e] = 0.0
        yr = oKr.apply_add(t.real, np.zeros(n)) - oKi.apply_add(t.imag, np.zeros(n))
        yi = oKi.apply_add(t.real, np.zeros(n)) + oKr.apply_add(t.imag, np.zeros(n))
        y = yr + 1j * yi
        y[e] = v[e]
        return y

    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    yr, yi = A.mult(_dev(x.real), _dev(x.imag), torch.empty(n, dtype=torch.float64, device="cuda"),
                    torch.empty(n, dtype=torch.float64, device="cuda"))
    y = yr.cpu().numpy() + 1j * yi.cpu().numpy()
    ref = oA(x)
    assert np.linalg.norm(y - ref) < 1e-12 * np.linalg.norm(ref)

    # preconditioner: Hiptmair p-multigrid on the shifted SPD matrix K + k0^2 eps M (real), applied to both parts
    shift = np.concatenate([coef(k0 ** 2 * eps), ident])
    pfine = nd_op(blocks[-1], ceed.QF_HDIVMASS_33, shift, ceed.EVAL_CURL | ceed.EVAL_INTERP)
    ploc = [pfine.coarsen_dense(b) for b in blocks[:-1]] + [pfine]
    Pm = [linalg.ParOperator(ctx, o, es, linalg.DIAG_ONE) for o, es in zip(ploc, ess)]
    h1s = [tet.H1TetSpace(mesh, q) for q in prob.orders]
    hb = [prob.h1_block(s) for s in h1s]
    hfine = ceed.Operator(h1s[-1].ndofs, h1s[-1].ndofs).add_dense_integrator(prob.geom, hb[-1], ceed.QF_HCURL_33,
                                                                            coef(k0 ** 2 * eps), ceed.EVAL_GRAD).finalize()
    hloc = [hfine.coarsen_dense(b) for b in hb[:-1]] + [hfine]
    Ph = [linalg.ParOperator(ctx, o, s.ess_dofs(fmask), linalg.DIAG_ONE) for o, s in zip(hloc, h1s)]
    G = [linalg.DenseInterp(ctx, h.restriction(), s.restriction(interp_range=True), tet.tet_gradient_matrix(q))
         for h, s, q in zip(h1s, prob.spaces, prob.orders)]
    P = [linalg.DenseInterp(ctx, prob.spaces[l].restriction(), prob.spaces[l + 1].restriction(interp_range=True),
                            tet.nd_tet_transfer_matrix(prob.orders[l], prob.orders[l + 1])) for l in range(len(Pm) - 1)]
    coarse = linalg.cg(ctx, Pm[0], linalg.jacobi(ctx, Pm[0]), rel_tol=1e-3, max_it=200)
    B = linalg.gmg(ctx, Pm, P, coarse, cheby_order=4, A_aux=Ph, G=G)

    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b[e] = 0.0
    S = linalg.ComplexParGmres(ctx, A, B, rel_tol=1e-8, max_it=400, restart=100, flexible=True)
    xr, xi = S.mult(_dev(b.real), _dev(b.imag), torch.zeros(n, dtype=torch.float64, device="cuda"),
                    torch.zeros(n, dtype=torch.float64, device="cuda"))
    st = S.stats()
    assert st["converged"], st
    xs = xr.cpu().numpy() + 1j * xi.cpu().numpy()
    res = np.linalg.norm(oA(xs) - b) / np.linalg.norm(b)
    assert res < 1e-6, (res, st)
    print(f"cpw p=3: {n} complex dofs, FGMRES + Hiptmair p-MG: {st['iterations']} iterations, oracle residual {res:.2e}")


def test_cpw_driven_solver_with_native_ams_and_refinement():
    """TetProblem.driven_solver (the configuration bench.py's cpw leg runs) on the reference mesh: the native AMS cycle on the
    assembled order-1 level inside the Hiptmair p-multigrid; the same system with the Jacobi-PCG coarse solve gives the same
    solution.  And one uniform refinement of the mesh (tet.refine_uniform) keeps volume, attributes and boundary tags."""
    from palace_amd import linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cpw_mesh.npz"))
    mesh = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
    fine = tet.refine_uniform(mesh)

    def vol(m, a):
        X = m.verts[m.tets[m.attr == a]]
        return np.einsum("ei,ei->e", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0]).sum() / 6

    assert fine.ne == 8 * mesh.ne and len(fine.bdr_tris) == 4 * len(mesh.bdr_tris)
    for a in (1, 2):
        assert abs(vol(fine, a) - vol(mesh, a)) < 1e-12 * vol(mesh, a)
    assert fine.boundary_face_mask.sum() == 4 * mesh.boundary_face_mask.sum()

    bt = np.sort(np.asarray(d["bdr_tris"], dtype=np.int64), axis=1)
    fkey = {tuple(f): i for i, f in enumerate(map(tuple, mesh.face_verts))}
    fmask = np.zeros(mesh.face_verts.shape[0], dtype=bool)
    fmask[[fkey[tuple(f)] for f in bt[np.isin(d["bdr_attr"], (4, 13))]]] = True
    k0 = 2 * np.pi * 16.0e9 * 1.0e-6 / 299792458.0
    sols, its = [], []
    for coarse in ("ams", "cg"):
        prob = TetProblem(linalg.Context(), mesh, 2)
        s = prob.driven_solver(fmask, k0, eps=[1.0, 11.7], tand=[0.0, 0.05], coarse=coarse, rel_tol=1e-10)
        n = s["n"]
        rng = np.random.default_rng(4)
        b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
        b[s["ess"]] = 0.0
        xr, xi = s["solver"].mult(_dev(b.real), _dev(b.imag), torch.zeros(n, dtype=torch.float64, device="cuda"),
                                  torch.zeros(n, dtype=torch.float64, device="cuda"))
        st = s["solver"].stats()
        assert st["converged"], (coarse, st)
        yr, yi = s["A"].mult(xr, xi, torch.empty_like(xr), torch.empty_like(xr))
        r = np.hypot(np.linalg.norm(yr.cpu().numpy() - b.real), np.linalg.norm(yi.cpu().numpy() - b.imag)) / np.linalg.norm(b)
        assert r < 1e-8, (coarse, r)
        sols.append(xr.cpu().numpy() + 1j * xi.cpu().numpy())
        its.append(st["iterations"])
    assert np.linalg.norm(sols[0] - sols[1]) < 1e-6 * np.linalg.norm(sols[1])
    assert its[0] <= 1.5 * its[1] + 5, its  # the one-cycle AMS coarse solve is not far from the converged PCG one
    print("cpw p=2 FGMRES iterations: AMS", its[0], " Jacobi-PCG(1e-3)", its[1])


# reference values: test/data/regression/ref/cpw/lumped_uniform/port-S.csv, first row (2 GHz), excitation 1:
# |S[j][1]| in dB and arg(S[j][1]) in degrees for j = 1..4 (the reference's regression gate: rtol 2e-2 per column)
PORT_S_2GHZ = {1: (-1.701793831224e+01, -1.148548267446e+02), 2: (-8.860531101837e-02, -2.478868071247e+01),
               3: (-5.250206957228e+01, +6.419829438631e+01), 4: (-6.287205660151e+01, +4.353831980561e+01)}


def test_cpw_lumped_uniform_as_the_reference_defines_it():
    """examples/cpw/cpw_lumped_uniform.json on the reference's mesh and order (14 628 tet4, ND p = 2): sapphire tensors, first-order
    absorbing boundary, four resistive lumped ports (surface f_apply_hcurl_32 terms in the imaginary part), PEC trace, uniform port
    excitation, 2 GHz.  (1) the device's complex operator against the oracle's volume and surface operators, 1e-12;
    (2) FGMRES + Hiptmair p-multigrid + AMS converges to 1e-8; (3) S[1..4][1] from the device solution against the one published pin
    of this configuration, port-S.csv (the reference gates each dB / degree column at rtol 2e-2)."""
    from oracle import palace_oracle as po
    from palace_amd import linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import CPW_LUMPED_UNIFORM, DrivenReferenceSystem, TetProblem

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cpw_mesh.npz"))
    mesh = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
    ctx = linalg.Context()
    prob = TetProblem(ctx, mesh, 2)
    ds = DrivenReferenceSystem(prob, 2.0, CPW_LUMPED_UNIFORM, rel_tol=1e-10, max_it=400)
    nd, n, k0 = ds.nd, ds.n, ds.k0
    # port geometry as lumpedelement.cpp derives it from the bounding boxes: 18 um squares, two elements per port
    for e in ds.port_elems.values():
        assert abs(e["width"] - 18.0) < 1e-9 and abs(e["length"] - 18.0) < 1e-9
        assert abs(e["Rs"] - 56.02 / 376.730313668 * 2.0) < 1e-12
    # ---- (1) operator parity: oracle volume operators (tensor coefficients) + oracle surface mass
    pts, wts = prob.pts, prob.wts
    interp, curl = nd.elem.tables(pts)
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    okw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    v = ds._vol
    o_r = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HDIVMASS,
                                po.CoeffCtx(attr_mat=v["amap"], mat_coeff=[-k0 ** 2 * m for m in v["eps"]]),
                                po.CoeffCtx(attr_mat=v["amap"], mat_coeff=v["mu_inv"]), **okw)
    o_iv = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HCURL,
                                 po.CoeffCtx(attr_mat=v["amap"], mat_coeff=[k0 ** 2 * m for m in v["eps_tand"]]), **okw)
    sint, scurl = ds.sblk.elem.tables(ds.spts)
    Js = ds.sblk.jacobians(ds.spts)
    ogs = po.build_geom_factor_32(ds.sblk.attr.astype(np.float64), ds.swts, np.transpose(Js, (0, 1, 3, 2)).reshape(ds.sblk.ne, -1, 6))
    o_is = po.CeedOperatorOracle(n, ds.sblk.offsets, ds.sblk.orients, sint, scurl, ogs, po.QF_HCURL_32,
                                 po.CoeffCtx(attr_mat=list(range(len(ds.scoef))), mat_coeff=[k0 * c for c in ds.scoef]))
    ess = ds.ess
    rng = np.random.default_rng(5)
    xr, xi = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)

    def o_complex(vr, vi):
        mr, mi = vr.copy(), vi.copy()
        mr[ess], mi[ess] = 0.0, 0.0
        z = np.zeros(n)
        ai = lambda w: o_iv.apply_add(w, z.copy()) + o_is.apply_add(w, z.copy())  # noqa: E731
        wr = o_r.apply_add(mr, z.copy()) - ai(mi)
        wi = o_r.apply_add(mi, z.copy()) + ai(mr)
        wr[ess], wi[ess] = vr[ess], vi[ess]
        return wr, wi

    wr, wi = o_complex(xr, xi)
    yr, yi = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    ds.A.mult(_dev(xr), _dev(xi), yr, yi)
    err = np.linalg.norm(np.concatenate([yr.cpu().numpy() - wr, yi.cpu().numpy() - wi])) / np.linalg.norm(np.concatenate([wr, wi]))
    assert err < 1e-12, err
    # ---- (2) the solve, excitation 1
    br, bi = ds.excitation(1)
    assert float(bi.abs().max()) > 0.0
    sr, si = torch.zeros_like(br), torch.zeros_like(br)
    ds.solver.mult(br, bi, sr, si)
    st = ds.solver.stats()
    assert st["converged"], st
    # the residual of the device solution in the ORACLE's operator
    ar, ai_ = o_complex(sr.cpu().numpy(), si.cpu().numpy())
    res = np.linalg.norm(np.concatenate([ar - br.cpu().numpy(), ai_ - bi.cpu().numpy()])) / float(torch.sqrt(bi @ bi))
    assert res < 1e-8, res
    # ---- (3) S-parameters against port-S.csv
    S = ds.s_parameters(sr, si, excited=1)
    got = {j: (20.0 * np.log10(abs(s)), np.degrees(np.angle(s))) for j, s in S.items()}
    print("S[j][1] (dB, deg):", got, "iterations", st["iterations"])
    for j, (db, deg) in PORT_S_2GHZ.items():
        assert abs(got[j][0] - db) <= 2.0e-2 * abs(db) + 1e-11, (j, got[j], (db, deg))
        dphi = (got[j][1] - deg + 180.0) % 360.0 - 180.0
        assert abs(dphi) <= 2.0e-2 * abs(deg) + 1e-11, (j, got[j], (db, deg))
