"""Generate tests/golden/qf_golden.npz by running the REFERENCE's own QFunction headers
(compiled into oracle/_ref by `make -C oracle ref`) on seeded random inputs.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
The fixtures pin oracle/palace_oracle.py and oracle/oracle_c.c on machines without the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from oracle import palace_oracle as po  # noqa: E402


def main():
    capi.build(ref=True)
    rng = np.random.default_rng(20260925)
    Q = 64
    # well-conditioned Jacobians: identity + perturbation, column-major [9][Q]
    J = (np.eye(3).reshape(9, 1) + 0.3 * rng.uniform(-1, 1, (9, Q))) * rng.uniform(0.5, 2.0, (1, Q))
    attr = rng.integers(1, 4, Q).astype(np.float64)  # 3 attributes
    qw = rng.uniform(0.01, 0.2, Q)
    geom = np.zeros((11, Q))
    capi.ref_call("f_build_geom_factor_33", None, Q, [attr, qw, np.ascontiguousarray(J)], [geom])
    # coefficient contexts: attr 1 -> anisotropic SPD, attr 2 -> scalar, attr 3 -> unassigned (zero)
    A = rng.uniform(-1, 1, (3, 3))
    aniso = A @ A.T + 3 * np.eye(3)
    ns = rng.uniform(-1, 1, (3, 3)) + 2 * np.eye(3)  # non-symmetric to expose transposition errors
    ctx_a = po.CoeffCtx(attr_mat=[0, 1, -1], mat_coeff=[aniso, np.array([2.08])])
    ctx_b = po.CoeffCtx(attr_mat=[1, 0, 1], mat_coeff=[ns, np.array([0.37])], a=1.5)
    ctx_id = po.CoeffCtx()
    u = rng.uniform(-1, 1, (3, Q))
    cu = rng.uniform(-1, 1, (3, Q))
    out = dict(Q=Q, J=J, attr=attr, qw=qw, geom=geom, u=u, cu=cu,
               ctx_a=ctx_a.pack(), ctx_b=ctx_b.pack(), ctx_id=ctx_id.pack(),
               ctx_pair=po.pack_pair(ctx_a, ctx_b))
    for tag, blob in (("a", ctx_a.pack()), ("b", ctx_b.pack()), ("id", ctx_id.pack())):
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hcurl_33", blob, Q, [geom, u], [v])
        out["hcurl_" + tag] = v
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hdiv_33", blob, Q, [geom, cu], [v])
        out["hdiv_" + tag] = v
    for tag, blob in (("a", ctx_a.pack()), ("b", ctx_b.pack())):  # the mixed H(curl) / H(div) QFunctions (hcurlhdiv_33_qf.h)
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hcurlhdiv_33", blob, Q, [geom, u], [v])
        out["hcurlhdiv_" + tag] = v
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hdivhcurl_33", blob, Q, [geom, cu], [v])
        out["hdivhcurl_" + tag] = v
    # the error QFunctions of the flux estimators (hcurlhdiv_error_33_qf.h): two inputs, pair context, one value per point
    for name in ("hcurlhdiv_error", "hdivhcurl_error"):
        w = np.zeros((1, Q))
        capi.ref_call("f_apply_%s_33" % name, po.pack_pair(ctx_a, ctx_b), Q, [geom, u, cu], [w])
        out[name] = w[0]
    v, cv = np.zeros((3, Q)), np.zeros((3, Q))
    capi.ref_call("f_apply_hdivmass_33", po.pack_pair(ctx_a, ctx_b), Q, [geom, u, cu], [v, cv])
    out["hdivmass_v"], out["hdivmass_cv"] = v, cv
    np.savez(os.path.join(ROOT, "tests", "golden", "qf_golden.npz"), **out)
    print("wrote qf_golden.npz")


if __name__ == "__main__":
    main()


def mesh_fixture():
    """The reference's own cylinder input (examples/cylinder/mesh/cylinder_hex.msh, 80 hex27)
    converted to arrays so GPU-box tests (no /root/reference there) run on the real mesh."""
    from palace_amd.fem.mesh import read_gmsh22

    m = read_gmsh22("/root/reference/examples/cylinder/mesh/cylinder_hex.msh")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cylinder_hex_mesh.npz"),
                        x=m.x, elem_nodes=m.elem_nodes.astype(np.int32), attr=m.attr,
                        bdr_faces=m.bdr_faces.astype(np.int32), bdr_attr=m.bdr_attr)
    print("wrote cylinder_hex_mesh.npz")


def fixtures_2d():
    """2-D QFunction vectors through the reference headers + the reference's cavity2d mesh and its
    regression eigenfrequencies (test/data/regression/ref/cavity2d/eigenmode/eig.csv)."""
    from palace_amd.fem import tri

    capi.build(ref=True)
    rng = np.random.default_rng(20260926)
    Q = 36
    J = (np.eye(2).reshape(4, 1) + 0.3 * rng.uniform(-1, 1, (4, Q))) * rng.uniform(0.5, 2.0, (1, Q))
    attr = rng.integers(1, 3, Q).astype(np.float64)
    qw = rng.uniform(0.01, 0.2, Q)
    geom = np.zeros((6, Q))
    capi.ref_call("f_build_geom_factor_22", None, Q, [attr, qw, np.ascontiguousarray(J)], [geom])
    A = rng.uniform(-1, 1, (2, 2))
    c2 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A @ A.T + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    u, cu = rng.uniform(-1, 1, (2, Q)), rng.uniform(-1, 1, (1, Q))
    out = dict(Q=Q, J=J, attr=attr, qw=qw, geom=geom, u=u, cu=cu, ctx2=c2.pack(), ctx1=c1.pack())
    v = np.zeros((2, Q))
    capi.ref_call("f_apply_hcurl_22", c2.pack(), Q, [geom, u], [v])
    out["hcurl_22"] = v.copy()
    w = np.zeros((1, Q))
    capi.ref_call("f_apply_l2_1", c1.pack(), Q, [geom, qw, cu], [w])
    out["l2_1"] = w.copy()
    pair = np.concatenate([c2.pack(), c1.pack()])
    v2, w2 = np.zeros((2, Q)), np.zeros((1, Q))
    capi.ref_call("f_apply_hdivmass_22", pair, Q, [geom, qw, u, cu], [v2, w2])
    out["hdivmass_22_v"], out["hdivmass_22_cv"], out["ctx_pair"] = v2, w2, pair
    # boundary elements (dim 2 in space_dim 3)
    J32 = rng.uniform(-1, 1, (6, Q)) + np.array([1, 0, 0, 0, 1, 0.3]).reshape(6, 1)
    g32 = np.zeros((8, Q))
    capi.ref_call("f_build_geom_factor_32", None, Q, [attr, qw, np.ascontiguousarray(J32)], [g32])
    B3 = rng.uniform(-1, 1, (3, 3))
    c3 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[B3 @ B3.T + 2 * np.eye(3), np.array([0.8])], a=0.9)
    v32 = np.zeros((2, Q))
    capi.ref_call("f_apply_hcurl_32", c3.pack(), Q, [g32, u], [v32])
    out.update(J32=J32, geom32=g32, ctx3=c3.pack(), hcurl_32=v32)
    # the remaining pair QFunctions in the plane and on boundary elements (no new random draws: earlier keys keep their bits)
    pair_h1 = np.concatenate([c1.pack(), c2.pack()])  # scalar mass first, then the 2x2 diffusion coefficient
    s22, g22 = np.zeros((1, Q)), np.zeros((2, Q))
    capi.ref_call("f_apply_hcurlmass_22", pair_h1, Q, [geom, cu, u], [s22, g22])
    pair_nd32 = np.concatenate([c3.pack(), c1.pack()])  # 3x3 mass first, then the scalar curl-curl coefficient
    v3, w3 = np.zeros((2, Q)), np.zeros((1, Q))
    capi.ref_call("f_apply_hdivmass_32", pair_nd32, Q, [g32, qw, u, cu], [v3, w3])
    pair_h132 = np.concatenate([c1.pack(), c3.pack()])
    s32, g32v = np.zeros((1, Q)), np.zeros((2, Q))
    capi.ref_call("f_apply_hcurlmass_32", pair_h132, Q, [g32, cu, u], [s32, g32v])
    out.update(hcurlmass_22_v=s22, hcurlmass_22_gv=g22, hdivmass_32_v=v3, hdivmass_32_cv=w3, hcurlmass_32_v=s32,
               hcurlmass_32_gv=g32v)
    # two-space QFunctions in the plane (hcurlhdiv_22_qf.h, hcurlhdiv_error_22_qf.h); new draws come last
    N2 = rng.uniform(-1, 1, (2, 2))
    c2n = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([0.7]), N2 + 2 * np.eye(2)], a=0.8, dim=2)  # non-symmetric
    u2 = rng.uniform(-1, 1, (2, Q))
    for name in ("hcurlhdiv_22", "hdivhcurl_22", "hdiv_22"):
        vv = np.zeros((2, Q))
        capi.ref_call("f_apply_" + name, c2n.pack(), Q, [geom, u], [vv])
        out[name] = vv
    pair22 = np.concatenate([c2n.pack(), c2.pack()])
    for name in ("hcurlhdiv_error_22", "hdivhcurl_error_22"):
        ee = np.zeros((1, Q))
        capi.ref_call("f_apply_" + name, pair22, Q, [geom, u, u2], [ee])
        out[name] = ee[0]
    # the scalar error integrand of the 2-D curl flux estimator (l2h1_error_qf.h), 1 x 1 pair context
    c1b = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([0.9]), np.array([1.3])], dim=1)
    cu2 = rng.uniform(-1, 1, (1, Q))
    ee = np.zeros((1, Q))
    capi.ref_call("f_apply_l2h1_error", np.concatenate([c1.pack(), c1b.pack()]), Q, [geom, cu, cu2], [ee])
    out.update(l2h1_error=ee[0], ctx1b=c1b.pack(), cu2=cu2)
    out.update(ctx2n=c2n.pack(), u2=u2)
    np.savez(os.path.join(ROOT, "tests", "golden", "qf2d_golden.npz"), **out)
    print("wrote qf2d_golden.npz")


def fixtures_line():
    """Line-element QFunction vectors (fem/qfunctions/21, 31) through the reference headers."""
    capi.build(ref=True)
    rng = np.random.default_rng(20260927)
    Q = 20
    attr = rng.integers(1, 3, Q).astype(np.float64)
    qw = rng.uniform(0.01, 0.2, Q)
    u, gu = rng.uniform(-1, 1, (1, Q)), rng.uniform(-1, 1, (1, Q))
    A2, A3 = rng.uniform(-1, 1, (2, 2)), rng.uniform(-1, 1, (3, 3))
    c2 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A2 + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)  # non-symmetric on purpose
    c3 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([0.8]), A3 + 2 * np.eye(3)], a=0.9)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    out = dict(Q=Q, attr=attr, qw=qw, u=u, gu=gu, ctx1=c1.pack(), ctx2=c2.pack(), ctx3=c3.pack())
    for sdim, cm in ((2, c2), (3, c3)):
        J = rng.uniform(-1, 1, (sdim, Q)) + np.array([1.0, 0.3, -0.2][:sdim]).reshape(sdim, 1)
        g = np.zeros((2 + sdim, Q))
        capi.ref_call("f_build_geom_factor_%d1" % sdim, None, Q, [attr, qw, np.ascontiguousarray(J)], [g])
        v = np.zeros((1, Q))
        capi.ref_call("f_apply_hcurl_%d1" % sdim, cm.pack(), Q, [g, u], [v])
        mv, gv = np.zeros((1, Q)), np.zeros((1, Q))
        capi.ref_call("f_apply_hcurlmass_%d1" % sdim, np.concatenate([c1.pack(), cm.pack()]), Q, [g, u, gu], [mv, gv])
        out.update({"J%d1" % sdim: J, "geom%d1" % sdim: g, "hcurl_%d1" % sdim: v, "hcurlmass_%d1_v" % sdim: mv,
                    "hcurlmass_%d1_gv" % sdim: gv})
    np.savez(os.path.join(ROOT, "tests", "golden", "qf1d_golden.npz"), **out)
    print("wrote qf1d_golden.npz")


def fixtures_rest():
    """The members of the QFunction families added last (round 4) through the reference headers: H(div) mass on boundary and line
    elements (hdiv_32 | _31 | _21), the mixed H(curl) / H(div) forms there (hcurlhdiv_32 | _31 | _21), div-div + mass
    (l2mass_22 | _33 | _32 | _31 | _21) and the gradient form (hcurlh1d_22 | _33 | _32 | _31 | _21), one geometry each."""
    capi.build(ref=True)
    rng = np.random.default_rng(20260928)
    Q = 24
    attr = rng.integers(1, 3, Q).astype(np.float64)
    qw = rng.uniform(0.01, 0.2, Q)
    A2, A3 = rng.uniform(-1, 1, (2, 2)), rng.uniform(-1, 1, (3, 3))
    c2 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A2 + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)  # non-symmetric on purpose
    c3 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([0.8]), A3 + 2 * np.eye(3)], a=0.9)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    out = dict(Q=Q, attr=attr, qw=qw, ctx1=c1.pack(), ctx2=c2.pack(), ctx3=c3.pack())
    base = {33: np.eye(3).reshape(9, 1), 22: np.eye(2).reshape(4, 1), 32: np.array([1, 0, 0, 0, 1, 0.3]).reshape(6, 1),
            31: np.array([1.0, 0.3, -0.2]).reshape(3, 1), 21: np.array([1.0, 0.3]).reshape(2, 1)}
    for tag in (33, 22, 32, 31, 21):
        sdim, dim = tag // 10, tag % 10
        cm = c3 if sdim == 3 else c2
        J = base[tag] + 0.4 * rng.uniform(-1, 1, (sdim * dim, Q))
        g = np.zeros((2 + sdim * dim, Q))
        capi.ref_call("f_build_geom_factor_%d" % tag, None, Q, [attr, qw, np.ascontiguousarray(J)], [g])
        u, du = rng.uniform(-1, 1, (dim, Q)), rng.uniform(-1, 1, (1, Q))
        out.update({"J%d" % tag: J, "geom%d" % tag: g, "u%d" % tag: u, "du%d" % tag: du})
        if tag in (32, 31, 21):
            for name in ("hdiv", "hcurlhdiv", "hdivhcurl"):
                v = np.zeros((dim, Q))
                capi.ref_call("f_apply_%s_%d" % (name, tag), cm.pack(), Q, [g, u], [v])
                out["%s_%d" % (name, tag)] = v
        v, dv = np.zeros((dim, Q)), np.zeros((1, Q))
        capi.ref_call("f_apply_l2mass_%d" % tag, np.concatenate([cm.pack(), c1.pack()]), Q, [g, qw, u, du], [v, dv])
        out.update({"l2mass_%d_v" % tag: v, "l2mass_%d_dv" % tag: dv})
        gv = np.zeros((sdim, Q))
        capi.ref_call("f_apply_hcurlh1d_%d" % tag, cm.pack(), Q, [g, u], [gv])
        out["hcurlh1d_%d" % tag] = gv
    # vector-valued scalar spaces (h1_2 | _3, l2_2 | _3: they read attr and w detJ only, any geometry data); new draws come last
    for n, cm, tag in ((2, c2, 22), (3, c3, 33)):
        uv = rng.uniform(-1, 1, (n, Q))
        g = out["geom%d" % tag]
        v, w = np.zeros((n, Q)), np.zeros((n, Q))
        capi.ref_call("f_apply_h1_%d" % n, cm.pack(), Q, [g, uv], [v])
        capi.ref_call("f_apply_l2_%d" % n, cm.pack(), Q, [g, qw, uv], [w])
        out.update({"uv%d" % n: uv, "h1_%d" % n: v, "l2_%d" % n: w})
    np.savez(os.path.join(ROOT, "tests", "golden", "qf_rest_golden.npz"), **out)
    print("wrote qf_rest_golden.npz")


def cavity2d_fixture():
    """The reference's cavity2d mesh and its regression values (eig.csv, terminal-M.csv, terminal-C.csv)."""
    from palace_amd.fem import tri

    m = tri.read_gmsh22_tris("/root/reference/examples/cavity2d/mesh/cavity2d.msh")
    eig = np.loadtxt("/root/reference/test/data/regression/ref/cavity2d/eigenmode/eig.csv", delimiter=",", skiprows=1)
    # boundary edges (vertex pairs in the node numbering of `nodes`) with their attributes, and the regression values of
    # the magnetostatic / electrostatic cases on the same mesh (terminal-M.csv, terminal-C.csv)
    used = np.unique(m.elem_nodes[:, :3])
    bdr = used[m.bdr_edges]
    ref = "/root/reference/test/data/regression/ref/cavity2d/"
    M11 = np.loadtxt(ref + "magnetostatic/terminal-M.csv", delimiter=",", skiprows=1)[1]
    C11 = np.loadtxt(ref + "electrostatic/terminal-C.csv", delimiter=",", skiprows=1)[1]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cavity2d_mesh.npz"), nodes=m.nodes,
                        elem_nodes=m.elem_nodes.astype(np.int32), attr=m.attr, eig_re_GHz=eig[:, 1], eig_im_GHz=eig[:, 2],
                        bdr_edges=bdr.astype(np.int32), bdr_attr=m.bdr_attr.astype(np.int32), M11_H=M11, C11_F=C11)
    print("wrote cavity2d_mesh.npz")


def spheres_fixture():
    """The reference's examples/spheres/mesh/spheres.msh (binary Gmsh 2.2, 14 362 tet20 + tri10 boundary faces) as arrays,
    with the element nodes re-ordered to palace_amd.fem.tet.h1_tet_nodes(3) (vertices, edges, faces; positive
    orientation), and the regression capacitance matrix test/data/regression/ref/spheres/terminal-C.csv."""
    import struct

    from palace_amd.fem import tet

    data = open("/root/reference/examples/spheres/mesh/spheres.msh", "rb").read()

    def section(name):
        a = data.index(b"$" + name + b"\n") + len(name) + 2
        return a, data.index(b"$End" + name)

    a, _ = section(b"Nodes")
    nl = data.index(b"\n", a)
    nn = int(data[a:nl])
    rec = np.frombuffer(data, dtype=np.dtype([("i", "<i4"), ("x", "<f8", 3)]), count=nn, offset=nl + 1)
    ids, xyz = rec["i"].astype(np.int64), rec["x"].copy()
    idmap = np.full(ids.max() + 1, -1, dtype=np.int64)
    idmap[ids] = np.arange(nn)
    a, _ = section(b"Elements")
    nl = data.index(b"\n", a)
    nelem = int(data[a:nl])
    npe = {29: 20, 21: 10, 15: 1, 26: 4}
    vol, vattr, tri, tattr = [], [], [], []
    off, done = nl + 1, 0
    while done < nelem:
        et, nf, nt = struct.unpack_from("<iii", data, off)
        off += 12
        w = 1 + nt + npe[et]
        r = np.frombuffer(data, dtype="<i4", count=w * nf, offset=off).reshape(nf, w)
        off += 4 * w * nf
        if et == 29:
            vol.append(r[:, 1 + nt:]), vattr.append(r[:, 1])
        elif et == 21:
            tri.append(r[:, 1 + nt:1 + nt + 3]), tattr.append(r[:, 1])
        done += nf
    en = idmap[np.concatenate(vol)]
    X = xyz[en]                                    # [ne, 20, 3]
    det = np.einsum("ei,ei->e", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
    # barycentric coordinates (numerators over 3) of every node wrt the element's own four vertices
    T = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2)      # columns
    lam = np.linalg.solve(T[:, None, :, :], (X - X[:, :1])[..., None])[..., 0]           # [ne, 20, 3] = (l1, l2, l3)
    # Gmsh orders the 20 nodes the same way in every element: read the lattice position of local node n off the
    # straight-sided elements (most of them), then apply it to all
    dev = np.abs(3 * lam - np.rint(3 * lam)).max(axis=(1, 2))
    straight = dev < 1e-6
    assert straight.sum() > 100
    k1 = np.rint(3 * lam[straight]).astype(np.int64)
    assert np.all(k1 == k1[:1]), "node ordering differs between elements"
    k = np.broadcast_to(k1[0], (en.shape[0], 20, 3)).copy()
    neg = det < 0
    k[neg] = k[neg][:, :, [1, 0, 2]]               # swapping vertices 1 and 2 swaps l1 and l2
    key = k[..., 0] + 4 * k[..., 1] + 16 * k[..., 2]
    std = np.rint(3 * tet.h1_tet_nodes(3)).astype(np.int64)     # reference coordinates = (l1, l2, l3)
    skey = std[:, 0] + 4 * std[:, 1] + 16 * std[:, 2]
    order = np.argsort(key, axis=1)
    assert np.array_equal(np.take_along_axis(key, order, axis=1), np.broadcast_to(np.sort(skey), key.shape))
    perm = np.empty_like(order)
    perm[:, np.argsort(skey)] = order              # perm[e, m] = node of element e at standard lattice point m
    en_std = np.take_along_axis(en, perm, axis=1)
    Xs = xyz[en_std]
    det2 = np.einsum("ei,ei->e", np.cross(Xs[:, 1] - Xs[:, 0], Xs[:, 2] - Xs[:, 0]), Xs[:, 3] - Xs[:, 0])
    assert det2.min() > 0
    C = np.loadtxt("/root/reference/test/data/regression/ref/spheres/terminal-C.csv", delimiter=",", skiprows=1)[:, 1:]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spheres_mesh.npz"), nodes=xyz,
                        elem_nodes=en_std.astype(np.int32), attr=np.concatenate(vattr).astype(np.int32),
                        bdr_tris=idmap[np.concatenate(tri)].astype(np.int32), bdr_attr=np.concatenate(tattr).astype(np.int32),
                        C_F=C)
    print("wrote spheres_mesh.npz", en_std.shape, int(neg.sum()), "re-oriented")


def cpw_fixture():
    """The reference's examples/cpw/mesh/cpw_lumped_0.msh (binary Gmsh 2.2: 14 628 tet4, boundary tri3 with the port /
    trace / far-field attributes) as arrays: vertices, positively oriented tetrahedra, volume attributes (1 air, 2 si,
    3 metal), boundary triangles and their attributes."""
    from palace_amd.fem import tet

    m = tet.read_gmsh22_tets("/root/reference/examples/cpw/mesh/cpw_lumped_0.msh")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cpw_mesh.npz"), verts=m.verts, tets=m.tets.astype(np.int32),
                        attr=m.attr.astype(np.int32), bdr_tris=np.asarray(m.bdr_tris, dtype=np.int32),
                        bdr_attr=np.asarray(m.bdr_attr, dtype=np.int32))
    print("wrote cpw_mesh.npz", m.tets.shape, m.verts.shape, np.unique(m.attr), np.unique(m.bdr_attr))


if __name__ == "__main__":
    mesh_fixture()
    fixtures_2d()
    fixtures_line()
    cavity2d_fixture()
    spheres_fixture()
    cpw_fixture()
