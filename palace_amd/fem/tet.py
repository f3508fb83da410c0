"""Tetrahedral meshes and finite element spaces (host-side set-up; plays the role MFEM plays for
Palace: it produces the tables and index arrays that cross the C ABI).

Reference behaviour restated here (MFEM is not in the reference tree):
  * Nedelec first-kind space of order p on the tetrahedron, nodal basis dual to tangential point
    functionals: p per edge, p(p-1) per face (pairs with the two face tangents), p(p-1)(p-2)/2
    interior (fem/libceed/basis.cpp:40-85 takes the dense tables of such an element);
  * element -> global dofs with edge signs and, for p >= 2, the 2x2 integer transformation of each
    face dof pair when the element sees the face with other vertices than the global face frame
    (MFEM's DofTransformation; fem/libceed/restriction.cpp:299-369 turns it into the tridiagonal
    int8 `curl_orients` this module also produces);
  * nodal H1 space of order p; order-1/2 nodal geometry.
The quadrature rule is the Stroud conical (collapsed Gauss-Jacobi) rule with p+1 points per direction
(exact to degree 2p+1 >= the reference's default order 2p, fem/integrator.cpp:14-39).
"""
from __future__ import annotations

import itertools
import struct

import numpy as np

from .basis1d import gauss_legendre

# reference tetrahedron: v0 = 0, v1 = e_x, v2 = e_y, v3 = e_z
REF_VERTS = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
LOCAL_EDGES = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
LOCAL_FACES = [(1, 2, 3), (0, 3, 2), (0, 1, 3), (0, 2, 1)]


# ---- quadrature --------------------------------------------------------------------------------

def tet_quadrature(n):
    """Conical product rule with n^3 points on the reference tetrahedron (weights sum to 1/6)."""
    from scipy.special import roots_jacobi

    t2, w2 = roots_jacobi(n, 2.0, 0.0)
    t1, w1 = roots_jacobi(n, 1.0, 0.0)
    t0, w0 = roots_jacobi(n, 0.0, 0.0)
    u, wu = 0.5 * (1 + t2), w2 / 8.0
    v, wv = 0.5 * (1 + t1), w1 / 4.0
    w, ww = 0.5 * (1 + t0), w0 / 2.0
    pts, wts = [], []
    for a in range(n):
        for b in range(n):
            for c in range(n):
                pts.append([u[a], v[b] * (1 - u[a]), w[c] * (1 - u[a]) * (1 - v[b])])
                wts.append(wu[a] * wv[b] * ww[c])
    return np.array(pts), np.array(wts)


def _orbit(bary):
    return np.array(sorted(set(itertools.permutations(bary))))


def tet_quadrature_symmetric(order):
    """Fully symmetric positive interior rules: 4 points (degree 2), 14 points (degree 5), 24 points
    (degree 6, Keast) — the point counts the reference gets from MFEM for orders 2 / 5 / 6
    ('Q = size of MFEM's order-2p tet rule').  The orbit parameters were obtained by
    solving the moment equations (tests/test_tet_space.py checks exactness); higher orders fall back to
    the conical product rule."""
    if order <= 2:
        a = (5.0 - np.sqrt(5.0)) / 20.0
        orbits = [((a, a, a, 1 - 3 * a), 1.0 / 24.0)]
    elif order <= 5:
        orbits = [((0.3108859192633006,) * 3 + (1 - 3 * 0.3108859192633006,), 0.01878132095300307),
                  ((0.09273525031089183,) * 3 + (1 - 3 * 0.09273525031089183,), 0.01224884051939384),
                  ((0.04550370412564658, 0.04550370412564658, 0.5 - 0.04550370412564658, 0.5 - 0.04550370412564658),
                   0.007091003462846512)]
    elif order <= 6:
        a1, a2, a3 = 0.32233789014228087, 0.04067395853460012, 0.21460287125920982
        b1, b2 = 0.06366100187502555, 0.2696723314583101
        orbits = [((a1, a1, a1, 1 - 3 * a1), 0.009226196923940297), ((a2, a2, a2, 1 - 3 * a2), 0.0016795351758862424),
                  ((a3, a3, a3, 1 - 3 * a3), 0.006653791709692282), ((b1, b1, b2, 1 - 2 * b1 - b2), 0.008035714285715949)]
    else:
        return tet_quadrature(order // 2 + 1)
    pts, wts = [], []
    for bary, w in orbits:
        o = _orbit(bary)
        pts.append(o[:, 1:])
        wts += [w] * len(o)
    wts = np.array(wts)
    return np.concatenate(pts), wts / wts.sum() / 6.0


def default_tet_rule(p):
    """Quadrature of order 2p, the reference default (fem/integrator.cpp:14-39)."""
    return tet_quadrature_symmetric(2 * p)


# ---- polynomials on the reference tet (monomials in centred coordinates) -------------------------

_CENTROID = np.array([0.25, 0.25, 0.25])


def _monomials(deg_lo, deg_hi):
    return [(a, b, c) for d in range(deg_lo, deg_hi + 1) for a in range(d + 1) for b in range(d + 1 - a)
            for c in [d - a - b]]


def _mono_eval(e, z):
    return z[..., 0] ** e[0] * z[..., 1] ** e[1] * z[..., 2] ** e[2]


def _mono_grad(e, z):
    g = np.zeros(z.shape)
    for d in range(3):
        if e[d] > 0:
            ee = list(e)
            ee[d] -= 1
            g[..., d] = e[d] * _mono_eval(ee, z)
    return g


class _VecPoly:
    """Vector polynomial given as a list of (component, exponent, coefficient)."""

    def __init__(self, terms):
        self.terms = terms

    def eval(self, x):
        z = x - _CENTROID
        v = np.zeros(z.shape)
        for c, e, a in self.terms:
            v[..., c] += a * _mono_eval(e, z)
        return v

    def curl(self, x):
        z = x - _CENTROID
        w = np.zeros(z.shape)
        for c, e, a in self.terms:
            g = a * _mono_grad(e, z)  # d f_c / d x_d
            # curl_i = eps_ijk d_j f_k : component c contributes to i with (j, k = c)
            i1, i2 = (c + 1) % 3, (c + 2) % 3
            # (curl)_{i1} = d_{i2} f_c ... sign bookkeeping via the Levi-Civita symbol
            w[..., i1] += g[..., i2]   # eps_{i1, i2, c} = +1
            w[..., i2] -= g[..., i1]   # eps_{i2, i1, c} = -1
        return w


def _nd_candidates(p):
    """A spanning set of the first-kind Nedelec space R_p = P_{p-1}^3 + z x P~_{p-1}^3 (z = centred
    coordinates; the space does not depend on the centre)."""
    cands = []
    for e in _monomials(0, p - 1):
        for c in range(3):
            cands.append(_VecPoly([(c, e, 1.0)]))
    for e in _monomials(p - 1, p - 1):
        for k in range(3):  # r = m e_k ; z x r = (z_j r_k - ...) : (z x r)_i = eps_ijk z_j r_k
            i1, i2 = (k + 1) % 3, (k + 2) % 3
            e1 = list(e)
            e1[i2] += 1
            e2 = list(e)
            e2[i1] += 1
            # (z x r)_{i1} = z_{i2} r_k * eps_{i1,i2,k} = + z_{i2} m ; (z x r)_{i2} = - z_{i1} m
            cands.append(_VecPoly([(i1, tuple(e1), 1.0), (i2, tuple(e2), -1.0)]))
    return cands


def _interior_lattice(dim, p_denominator, n_sum):
    """Barycentric lattice points (i_0+1, ..., i_dim+1) / p_denominator with sum(i) = n_sum."""
    out = []
    for idx in itertools.product(range(n_sum + 1), repeat=dim):
        last = n_sum - sum(idx)
        if last >= 0:
            out.append(tuple((i + 1) / p_denominator for i in idx + (last,)))
    return out


def face_point_bary(p):
    """Barycentric coordinates (wrt the face's three vertices) of the p(p-1)/2 face dof points."""
    return _interior_lattice(2, p + 1, p - 2) if p >= 2 else []


def nd_tet_functionals(p):
    """Dof functionals (point, tangent) of the order-p Nedelec tet in local order: edges, faces, interior."""
    V = REF_VERTS
    eo = gauss_legendre(p)[0]  # open points in (0, 1)
    pts, tans = [], []
    for a, b in LOCAL_EDGES:
        for t in eo:
            pts.append((1 - t) * V[a] + t * V[b])
            tans.append(V[b] - V[a])
    for a, b, c in LOCAL_FACES:
        for la, lb, lc in face_point_bary(p):
            x = la * V[a] + lb * V[b] + lc * V[c]
            pts += [x, x]
            tans += [V[b] - V[a], V[c] - V[a]]
    if p >= 3:
        for l in _interior_lattice(3, p + 1, p - 3):
            x = sum(li * V[i] for i, li in enumerate(l))
            for d in range(3):
                pts.append(x)
                tans.append(np.eye(3)[d])
    return np.array(pts), np.array(tans)


class NDTetElement:
    """Order-p Nedelec tetrahedron: tables at arbitrary reference points."""

    def __init__(self, p):
        self.p = p
        self.P = p * (p + 2) * (p + 3) // 2
        self.cands = _nd_candidates(p)
        pts, tans = nd_tet_functionals(p)
        assert len(pts) == self.P
        self.dof_pts, self.dof_tans = pts, tans
        Vm = np.array([np.einsum("nd,nd->n", c.eval(pts), tans) for c in self.cands]).T  # [P, ncand]
        self.coef = np.linalg.pinv(Vm, rcond=1e-12)  # [ncand, P]; basis_j = sum_k coef[k, j] cand_k
        assert np.abs(Vm @ self.coef - np.eye(self.P)).max() < 1e-9, "Nedelec dofs are not unisolvent"

    def tables(self, x):
        """interp [3, Q, P], curl [3, Q, P] at reference points x [Q, 3]."""
        val = np.array([c.eval(x) for c in self.cands])   # [ncand, Q, 3]
        cur = np.array([c.curl(x) for c in self.cands])
        interp = np.einsum("kqd,kj->dqj", val, self.coef)
        curl = np.einsum("kqd,kj->dqj", cur, self.coef)
        return np.ascontiguousarray(interp), np.ascontiguousarray(curl)


def h1_tet_nodes(p):
    """Equispaced order-p lattice in local order: vertices, edges, faces, interior."""
    V = REF_VERTS
    pts = [V[i] for i in range(4)]
    for a, b in LOCAL_EDGES:
        for i in range(1, p):
            pts.append(V[a] + (V[b] - V[a]) * i / p)
    for a, b, c in LOCAL_FACES:
        for la, lb, lc in (_interior_lattice(2, p, p - 3) if p >= 3 else []):
            pts.append(la * V[a] + lb * V[b] + lc * V[c])
    for l in (_interior_lattice(3, p, p - 4) if p >= 4 else []):
        pts.append(sum(li * V[i] for i, li in enumerate(l)))
    return np.array(pts)


class H1TetElement:
    def __init__(self, p):
        self.p = p
        self.P = (p + 1) * (p + 2) * (p + 3) // 6
        self.nodes = h1_tet_nodes(p)
        assert len(self.nodes) == self.P
        self.monos = _monomials(0, p)
        Vm = np.array([_mono_eval(e, self.nodes - _CENTROID) for e in self.monos]).T
        self.coef = np.linalg.inv(Vm)

    def tables(self, x):
        z = x - _CENTROID
        val = np.array([_mono_eval(e, z) for e in self.monos])       # [k, Q]
        grd = np.array([_mono_grad(e, z) for e in self.monos])       # [k, Q, 3]
        interp = np.einsum("kq,kj->qj", val, self.coef)[None]
        grad = np.einsum("kqd,kj->dqj", grd, self.coef)
        return np.ascontiguousarray(interp), np.ascontiguousarray(grad)


# ---- mesh ----------------------------------------------------------------------------------------

class TetMesh:
    """Conforming tetrahedral mesh.  tets [ne, 4] vertex ids; geometry nodes of order 1 (the vertices)
    or 2 (tet10: vertices + edge midpoints in LOCAL_EDGES order)."""

    def __init__(self, verts, tets, attr=None, elem_nodes=None, nodes=None, bdr_tris=None, bdr_attr=None):
        self.verts = np.asarray(verts, dtype=np.float64)
        self.tets = np.asarray(tets, dtype=np.int64)
        self.ne = self.tets.shape[0]
        self.attr = np.ones(self.ne, dtype=np.int32) if attr is None else np.asarray(attr, dtype=np.int32)
        self.nodes = self.verts if nodes is None else np.asarray(nodes, dtype=np.float64)
        self.elem_nodes = self.tets if elem_nodes is None else np.asarray(elem_nodes, dtype=np.int64)
        self.mesh_order = 1 if self.elem_nodes.shape[1] == 4 else 2
        self.bdr_tris, self.bdr_attr = bdr_tris, bdr_attr
        # positive orientation
        X = self.verts[self.tets]
        det = np.einsum("ei,ei->e", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
        if np.any(det <= 0):
            raise ValueError("inverted tetrahedra")
        self._build_topology()

    def _build_topology(self):
        t = self.tets
        e = np.stack([np.sort(t[:, list(le)], axis=1) for le in LOCAL_EDGES], axis=1)   # [ne, 6, 2]
        ue, inv = np.unique(e.reshape(-1, 2), axis=0, return_inverse=True)
        self.edge_verts, self.elem_edges = ue, inv.reshape(self.ne, 6)
        f = np.stack([np.sort(t[:, list(lf)], axis=1) for lf in LOCAL_FACES], axis=1)   # [ne, 4, 3]
        uf, inv, cnt = np.unique(f.reshape(-1, 3), axis=0, return_inverse=True, return_counts=True)
        self.face_verts, self.elem_faces = uf, inv.reshape(self.ne, 4)
        self.boundary_face_mask = cnt == 1
        if np.any(cnt > 2):
            raise ValueError("non-manifold mesh")

    @property
    def nv(self):
        return self.verts.shape[0]

    def geometry_grad_table(self, x):
        """d phi_n / d xi_d of the nodal geometry basis at reference points x: [3, Q, npe]."""
        l = np.stack([1 - x.sum(axis=1), x[:, 0], x[:, 1], x[:, 2]], axis=1)   # [Q, 4]
        dl = np.array([[-1.0, -1, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]])       # [4, 3]
        Q = x.shape[0]
        if self.mesh_order == 1:
            return np.ascontiguousarray(np.broadcast_to(dl.T[:, None, :], (3, Q, 4)))
        G = np.zeros((3, Q, 10))
        for i in range(4):
            G[:, :, i] = ((4 * l[:, i] - 1)[None, :]) * dl[i][:, None]
        for k, (a, b) in enumerate(LOCAL_EDGES):
            G[:, :, 4 + k] = 4 * (l[:, a][None, :] * dl[b][:, None] + l[:, b][None, :] * dl[a][:, None])
        return G

    def jacobians(self, x):
        """J[e, q, i, d] = d x_i / d xi_d."""
        G = self.geometry_grad_table(x)
        return np.einsum("dqn,eni->eqid", G, self.nodes[self.elem_nodes])


def refine_uniform(mesh: "TetMesh") -> "TetMesh":
    """One level of uniform (red) refinement of a straight-sided mesh: every tetrahedron into 8 (the 4 corner children and the
    inner octahedron cut along the diagonal m02-m13: Bey's refinement, what mfem::Mesh::UniformRefinement does to tetrahedra
    up to the choice of the diagonal), every boundary triangle into 4; attributes are inherited."""
    if mesh.mesh_order != 1:
        raise ValueError("uniform refinement is implemented for straight-sided tetrahedra")
    nv = mesh.nv
    mid = nv + mesh.elem_edges  # [ne, 6] vertex number of the midpoint of local edge k (LOCAL_EDGES order)
    verts = np.concatenate([mesh.verts, 0.5 * (mesh.verts[mesh.edge_verts[:, 0]] + mesh.verts[mesh.edge_verts[:, 1]])])
    ek = {tuple(e): k for k, e in enumerate(LOCAL_EDGES)}
    m = lambda a, b: mid[:, ek[(min(a, b), max(a, b))]]
    v = [mesh.tets[:, i] for i in range(4)]
    m01, m02, m03, m12, m13, m23 = m(0, 1), m(0, 2), m(0, 3), m(1, 2), m(1, 3), m(2, 3)
    kids = [(v[0], m01, m02, m03), (m01, v[1], m12, m13), (m02, m12, v[2], m23), (m03, m13, m23, v[3]),
            (m01, m02, m03, m13), (m01, m02, m12, m13), (m02, m03, m13, m23), (m02, m12, m13, m23)]
    tets = np.stack([np.stack(k, axis=1) for k in kids], axis=1).reshape(-1, 4)
    X = verts[tets]
    det = np.einsum("ei,ei->e", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
    neg = det < 0
    tets[neg] = tets[neg][:, [0, 2, 1, 3]]
    attr = np.repeat(mesh.attr, 8)
    bt = ba = None
    if mesh.bdr_tris is not None:
        b = np.asarray(mesh.bdr_tris, dtype=np.int64)
        eid = {tuple(e): i for i, e in enumerate(map(tuple, mesh.edge_verts))}
        bm = lambda i, j: np.array([nv + eid[(min(x, y), max(x, y))] for x, y in zip(b[:, i], b[:, j])], dtype=np.int64)
        a01, a12, a02 = bm(0, 1), bm(1, 2), bm(0, 2)
        bt = np.stack([np.stack(t, axis=1) for t in ((b[:, 0], a01, a02), (a01, b[:, 1], a12), (a02, a12, b[:, 2]),
                                                     (a01, a12, a02))], axis=1).reshape(-1, 3)
        ba = None if mesh.bdr_attr is None else np.repeat(np.asarray(mesh.bdr_attr), 4)
    return TetMesh(verts, tets, attr, bdr_tris=bt, bdr_attr=ba)


def cube_tet_mesh(n, L=1.0, attr=None):
    """n^3 cubes of [0, L]^3, each split into the 6 Kuhn tetrahedra (conforming)."""
    g = np.linspace(0.0, L, n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    vid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k
    tets = []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                for perm in itertools.permutations(range(3)):
                    c = [i, j, k]
                    path = [vid(*c)]
                    for d in perm:
                        c[d] += 1
                        path.append(vid(*c))
                    tets.append(path)
    tets = np.array(tets)
    Xv = verts[tets]
    det = np.einsum("ei,ei->e", np.cross(Xv[:, 1] - Xv[:, 0], Xv[:, 2] - Xv[:, 0]), Xv[:, 3] - Xv[:, 0])
    neg = det < 0
    tets[neg] = tets[neg][:, [0, 2, 1, 3]]
    return TetMesh(verts, tets, attr)


def hex_to_tets(hexmesh, quadratic=False):
    """Split every hex27 into 24 tetrahedra around its body-centre node (face centre, body centre, one
    face edge): conforming without any diagonal choice.  Straight-sided unless `quadratic`."""
    lat = lambda i, j, k: i + 3 * (j + 3 * k)
    faces = []  # (centre node, 4 corner nodes in cyclic order)
    for d in range(3):
        for s in (0, 2):
            idx = [0, 0, 0]
            idx[d] = s
            o = [a for a in range(3) if a != d]
            cyc = []
            for u, v in ((0, 0), (2, 0), (2, 2), (0, 2)):
                idx[o[0]], idx[o[1]] = u, v
                cyc.append(lat(*idx))
            idx[o[0]], idx[o[1]] = 1, 1
            faces.append((lat(*idx), cyc))
    body = lat(1, 1, 1)
    tets = []
    for fc, cyc in faces:
        for a in range(4):
            tets.append([body, fc, cyc[a], cyc[(a + 1) % 4]])
    loc = np.array(tets)                                    # [24, 4] lattice node ids
    t = hexmesh.elem_nodes[:, loc].reshape(-1, 4)           # global node ids
    used, inv = np.unique(t, return_inverse=True)
    verts = hexmesh.x[used]
    t = inv.reshape(-1, 4)
    Xv = verts[t]
    det = np.einsum("ei,ei->e", np.cross(Xv[:, 1] - Xv[:, 0], Xv[:, 2] - Xv[:, 0]), Xv[:, 3] - Xv[:, 0])
    neg = det < 0
    t[neg] = t[neg][:, [0, 2, 1, 3]]
    attr = np.repeat(hexmesh.attr, 24)
    return TetMesh(verts, t, attr)


def read_gmsh22_tets(path):
    """Gmsh 2.2 (binary or ASCII) with tet4 (type 4) or tet10 (type 11) elements and tri3 / tri6
    boundary elements — the format of the reference's examples/*/mesh/*_tet.msh files."""
    data = open(path, "rb").read()

    def section(name):
        a = data.index(b"$" + name + b"\n") + len(name) + 2
        return a, data.index(b"$End" + name)

    a, _ = section(b"MeshFormat")
    binary = int(data[a : data.index(b"\n", a)].split()[1]) == 1
    a, b = section(b"Nodes")
    nl = data.index(b"\n", a)
    nn = int(data[a:nl])
    ids, xyz = np.empty(nn, dtype=np.int64), np.empty((nn, 3))
    if binary:
        rec = np.frombuffer(data, dtype=np.dtype([("i", "<i4"), ("x", "<f8", 3)]), count=nn, offset=nl + 1)
        ids[:], xyz[:] = rec["i"], rec["x"]
    else:
        rows = data[nl + 1 : b].split()
        for n in range(nn):
            ids[n] = int(rows[4 * n])
            xyz[n] = [float(v) for v in rows[4 * n + 1 : 4 * n + 4]]
    idmap = np.full(ids.max() + 1, -1, dtype=np.int64)
    idmap[ids] = np.arange(nn)
    a, b = section(b"Elements")
    nl = data.index(b"\n", a)
    nelem = int(data[a:nl])
    nnodes = {4: 4, 11: 10, 2: 3, 9: 6, 15: 1, 1: 2, 8: 3}
    vol, vattr, tri, tattr = [], [], [], []
    if binary:
        off, done = nl + 1, 0
        while done < nelem:
            etype, nfollow, ntags = struct.unpack_from("<iii", data, off)
            off += 12
            npe = nnodes[etype]
            w = 1 + ntags + npe
            rec = np.frombuffer(data, dtype="<i4", count=w * nfollow, offset=off).reshape(nfollow, w)
            off += 4 * w * nfollow
            if etype in (4, 11):
                vol.append(rec[:, 1 + ntags :]), vattr.append(rec[:, 1])
            elif etype in (2, 9):
                tri.append(rec[:, 1 + ntags :]), tattr.append(rec[:, 1])
            done += nfollow
    else:
        for line in data[nl + 1 : b].splitlines():
            rec = [int(v) for v in line.split()]
            if not rec:
                continue
            etype, ntags = rec[1], rec[2]
            if etype in (4, 11):
                vol.append(np.array([rec[3 + ntags :]])), vattr.append(np.array([rec[3]]))
            elif etype in (2, 9):
                tri.append(np.array([rec[3 + ntags :]])), tattr.append(np.array([rec[3]]))
    if not vol:
        raise ValueError("no tetrahedra in " + path)
    en = idmap[np.concatenate(vol)]
    # gmsh tet10 edge order (0,1),(1,2),(0,2),(0,3),(2,3),(1,3) -> LOCAL_EDGES order
    if en.shape[1] == 10:
        en = en[:, [0, 1, 2, 3, 4, 6, 7, 5, 9, 8]]
    t = en[:, :4].copy()
    X = xyz[t]
    det = np.einsum("ei,ei->e", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
    neg = det < 0
    if np.any(neg):
        sw = [0, 2, 1, 3] if en.shape[1] == 4 else [0, 2, 1, 3, 5, 4, 6, 7, 9, 8]
        en[neg] = en[neg][:, sw]
        t = en[:, :4].copy()
    used, inv = np.unique(t, return_inverse=True)
    vmap = np.full(nn, -1, dtype=np.int64)
    vmap[used] = np.arange(used.size)
    bt = vmap[idmap[np.concatenate(tri)][:, :3]] if tri else None
    return TetMesh(xyz[used], inv.reshape(-1, 4), np.concatenate(vattr), elem_nodes=en if en.shape[1] == 10 else None,
                   nodes=xyz if en.shape[1] == 10 else None, bdr_tris=bt,
                   bdr_attr=np.concatenate(tattr) if tri else None)


# ---- spaces --------------------------------------------------------------------------------------

class NDTetSpace:
    """Order-p Nedelec space on a TetMesh: global dofs = edges | faces | interiors.

    offsets [ne, P] and curl_orients [ne, P, 3] (int8 rows {sub, main, super} of the element's
    tridiagonal transformation, signs folded in) are what fem/libceed/restriction.cpp:299-369 passes
    to CeedElemRestrictionCreateCurlOriented; for p = 1 the transformation is diagonal and `orients`
    (bool) is the oriented form (:288-298)."""

    def __init__(self, mesh: TetMesh, p: int):
        self.mesh, self.p = mesh, p
        self.elem = NDTetElement(p)
        self.P = self.elem.P
        ne = mesh.ne
        n_e, n_f, n_i = p, p * (p - 1), p * (p - 1) * (p - 2) // 2
        NE_, NF_ = mesh.edge_verts.shape[0], mesh.face_verts.shape[0]
        self.ndofs = NE_ * n_e + NF_ * n_f + ne * n_i
        self.edge_base, self.face_base, self.int_base = 0, NE_ * n_e, NE_ * n_e + NF_ * n_f
        off = np.zeros((ne, self.P), dtype=np.int64)
        T = np.zeros((ne, self.P, 3), dtype=np.int8)
        t = mesh.tets
        # edges
        for k, (a, b) in enumerate(LOCAL_EDGES):
            flip = t[:, a] > t[:, b]
            ge = mesh.elem_edges[:, k]
            for i in range(p):
                gi = np.where(flip, p - 1 - i, i)
                off[:, k * p + i] = ge * n_e + gi
                T[:, k * p + i, 1] = np.where(flip, -1, 1)
        # faces
        fb = face_point_bary(p)
        npt = len(fb)
        key = {tuple(np.round(np.array(l) * (p + 1)).astype(int)): m for m, l in enumerate(fb)}
        base = 6 * p
        for k, lf in enumerate(LOCAL_FACES):
            gv = t[:, list(lf)]                            # global ids of local (A, B, C)
            rank = np.argsort(np.argsort(gv, axis=1), axis=1)   # rank[e, m] = position of local vertex m in the sorted frame
            gf = mesh.elem_faces[:, k]
            # frame vectors of the sorted (global) face: e(0) = 0, e(1) = T1, e(2) = T2
            ev = np.array([[0, 0], [1, 0], [0, 1]])
            for m, l in enumerate(fb):
                li = np.round(np.array(l) * (p + 1)).astype(int)   # local barycentric numerators (A, B, C)
                # global barycentric: coordinate of sorted vertex s is the local one of the vertex with rank s
                gl = np.zeros((ne, 3), dtype=int)
                for mm in range(3):
                    gl[np.arange(ne), rank[:, mm]] = li[mm]
                gm = np.array([key[tuple(r)] for r in gl])
                # local tangents t1 = B - A, t2 = C - A in the global frame
                M1 = ev[rank[:, 1]] - ev[rank[:, 0]]       # [ne, 2] coefficients of (T1, T2)
                M2 = ev[rank[:, 2]] - ev[rank[:, 0]]
                j0 = base + k * n_f + 2 * m
                off[:, j0] = self.face_base + gf * n_f + 2 * gm
                off[:, j0 + 1] = self.face_base + gf * n_f + 2 * gm + 1
                T[:, j0, 1], T[:, j0, 2] = M1[:, 0], M1[:, 1]
                T[:, j0 + 1, 0], T[:, j0 + 1, 1] = M2[:, 0], M2[:, 1]
        # interior
        ib = base + 4 * n_f
        for i in range(n_i):
            off[:, ib + i] = self.int_base + np.arange(ne) * n_i + i
            T[:, ib + i, 1] = 1
        self.offsets = off.astype(np.int32)
        self.curl_orients = T
        self.diagonal_transform = bool(np.all(T[:, :, 0] == 0) and np.all(T[:, :, 2] == 0))
        self.orients = (T[:, :, 1] < 0) if self.diagonal_transform else None

    def restriction(self, interp_range=False):
        """dict(offsets, lsize, orients | curl_orients) as the C ABI takes it.  interp_range: the form
        Palace builds for the RANGE of an interpolator (InvTransformDual, restriction.cpp:318-336): rows of
        B with E_range^T = B^T = T^-1, i.e. B = T^-T, again 2x2 integer blocks."""
        r = dict(offsets=self.offsets, lsize=self.ndofs)
        if self.diagonal_transform:
            r["orients"] = self.orients  # +-1: its own inverse
            return r
        if not interp_range:
            r["curl_orients"] = self.curl_orients
            return r
        T = self.curl_orients.astype(np.int64)
        B = np.zeros_like(T)
        B[:, :, 1] = T[:, :, 1]  # diagonal (+-1) entries: own inverse
        p = self.p
        n_f = p * (p - 1)
        base = 6 * p
        for j0 in range(base, base + 4 * n_f, 2):
            a, b, c, d = T[:, j0, 1], T[:, j0, 2], T[:, j0 + 1, 0], T[:, j0 + 1, 1]
            det = a * d - b * c
            assert np.all(np.abs(det) == 1)
            B[:, j0, 1], B[:, j0, 2] = d * det, -c * det          # rows of T^-T = (1/det) [[d, -c], [-b, a]]
            B[:, j0 + 1, 0], B[:, j0 + 1, 1] = -b * det, a * det
        r["curl_orients"] = B.astype(np.int8)
        return r

    def ess_dofs(self, face_mask=None):
        """Dofs on boundary faces (all of them, or those selected by face_mask over mesh.face_verts)."""
        m = self.mesh
        fm = m.boundary_face_mask if face_mask is None else face_mask
        p = self.p
        n_e, n_f = p, p * (p - 1)
        faces = np.nonzero(fm)[0]
        fv = m.face_verts[faces]
        ekey = {tuple(e): i for i, e in enumerate(map(tuple, m.edge_verts))}
        edges = set()
        for f in fv:
            for a, b in ((0, 1), (0, 2), (1, 2)):
                edges.add(ekey[(f[a], f[b])])
        edges = np.array(sorted(edges), dtype=np.int64)
        d = [(edges[:, None] * n_e + np.arange(n_e)[None, :]).ravel(),
             (self.face_base + faces[:, None] * n_f + np.arange(n_f)[None, :]).ravel()]
        return np.unique(np.concatenate(d)).astype(np.int32)

    def interpolate(self, F):
        """Nodal interpolant of a smooth field F(x) -> [..., 3]: global dof = F(x) . t over the GLOBAL
        frame; computed per element through the inverse of the element transformation."""
        m, el = self.mesh, self.elem
        J = m.jacobians(el.dof_pts)                                # [ne, P, 3, 3]
        l = np.stack([1 - el.dof_pts.sum(axis=1), *el.dof_pts.T], axis=1)
        G = m.geometry_grad_table(el.dof_pts)
        # physical points through the geometry basis values
        if m.mesh_order == 1:
            X = np.einsum("qn,eni->eqi", l, m.nodes[m.elem_nodes])
        else:
            phi = np.concatenate([l * (2 * l - 1), np.stack([4 * l[:, a] * l[:, b] for a, b in LOCAL_EDGES], axis=1)], axis=1)
            X = np.einsum("qn,eni->eqi", phi, m.nodes[m.elem_nodes])
        tphys = np.einsum("eqid,qd->eqi", J, el.dof_tans)
        loc = np.einsum("eqi,eqi->eq", F(X), tphys)                # local functionals
        # local = T x_e  ->  x_e = T^{-1} local (2x2 blocks)
        x = np.zeros(self.ndofs)
        ne, P = loc.shape
        Tm = np.zeros((ne, P, P))
        r = np.arange(P)
        T = self.curl_orients.astype(np.float64)
        Tm[:, r, r] = T[:, :, 1]
        Tm[:, r[1:], r[:-1]] = T[:, 1:, 0]
        Tm[:, r[:-1], r[1:]] = T[:, :-1, 2]
        xe = np.linalg.solve(Tm, loc[:, :, None])[:, :, 0]
        x[self.offsets.ravel()] = xe.ravel()
        return x


def nd_tet_transfer_matrix(pc, pf):
    """Element matrix of the p-prolongation ND(pc) -> ND(pf) (MFEM GetTransferMatrix, basis.cpp:132-138):
    fine dof functionals applied to the coarse basis, [P_f, P_c]."""
    fine = NDTetElement(pf)
    interp, _ = NDTetElement(pc).tables(fine.dof_pts)      # [3, P_f, P_c]
    return np.ascontiguousarray(np.einsum("dji,jd->ji", interp, fine.dof_tans))


def h1_tet_transfer_matrix(pc, pf):
    interp, _ = H1TetElement(pc).tables(h1_tet_nodes(pf))  # [1, P_f, P_c]
    return np.ascontiguousarray(interp[0])


def tet_gradient_matrix(p):
    """Element matrix of the discrete gradient H1(p) -> ND(p) (basis.cpp:139-143): ND functionals of the
    gradients of the H1 basis, [P_nd, P_h1]."""
    nd = NDTetElement(p)
    _, grad = H1TetElement(p).tables(nd.dof_pts)           # [3, P_nd, P_h1]
    return np.ascontiguousarray(np.einsum("dji,jd->ji", grad, nd.dof_tans))


class H1TetSpace:
    """Order-p nodal H1 space on a TetMesh: global dofs = vertices | edges | faces | interiors."""

    def restriction(self, interp_range=False):
        return dict(offsets=self.offsets, lsize=self.ndofs)

    def __init__(self, mesh: TetMesh, p: int):
        self.mesh, self.p = mesh, p
        self.elem = H1TetElement(p)
        self.P = self.elem.P
        ne, t = mesh.ne, mesh.tets
        n_e, n_f, n_i = p - 1, (p - 1) * (p - 2) // 2, (p - 1) * (p - 2) * (p - 3) // 6
        NE_, NF_ = mesh.edge_verts.shape[0], mesh.face_verts.shape[0]
        self.edge_base = mesh.nv
        self.face_base = self.edge_base + NE_ * n_e
        self.int_base = self.face_base + NF_ * n_f
        self.ndofs = self.int_base + ne * n_i
        off = np.zeros((ne, self.P), dtype=np.int64)
        off[:, :4] = t
        for k, (a, b) in enumerate(LOCAL_EDGES):
            flip = t[:, a] > t[:, b]
            for i in range(n_e):
                off[:, 4 + k * n_e + i] = self.edge_base + mesh.elem_edges[:, k] * n_e + np.where(flip, n_e - 1 - i, i)
        fb = _interior_lattice(2, p, p - 3) if p >= 3 else []
        key = {tuple(np.round(np.array(l) * p).astype(int)): m for m, l in enumerate(fb)}
        base = 4 + 6 * n_e
        for k, lf in enumerate(LOCAL_FACES):
            gv = t[:, list(lf)]
            rank = np.argsort(np.argsort(gv, axis=1), axis=1)
            for m, l in enumerate(fb):
                li = np.round(np.array(l) * p).astype(int)
                gl = np.zeros((ne, 3), dtype=int)
                for mm in range(3):
                    gl[np.arange(ne), rank[:, mm]] = li[mm]
                gm = np.array([key[tuple(r)] for r in gl])
                off[:, base + k * n_f + m] = self.face_base + mesh.elem_faces[:, k] * n_f + gm
        for i in range(n_i):
            off[:, base + 4 * n_f + i] = self.int_base + np.arange(ne) * n_i + i
        self.offsets = off.astype(np.int32)

    def ess_dofs(self, face_mask=None):
        m, p = self.mesh, self.p
        fm = m.boundary_face_mask if face_mask is None else face_mask
        n_e, n_f = p - 1, (p - 1) * (p - 2) // 2
        faces = np.nonzero(fm)[0]
        fv = m.face_verts[faces]
        ekey = {tuple(e): i for i, e in enumerate(map(tuple, m.edge_verts))}
        edges = set()
        for f in fv:
            for a, b in ((0, 1), (0, 2), (1, 2)):
                edges.add(ekey[(f[a], f[b])])
        edges = np.array(sorted(edges), dtype=np.int64)
        d = [np.unique(fv.ravel()),
             (self.edge_base + edges[:, None] * n_e + np.arange(n_e)[None, :]).ravel(),
             (self.face_base + faces[:, None] * n_f + np.arange(n_f)[None, :]).ravel()]
        return np.unique(np.concatenate(d)).astype(np.int32)


def lowest_order_gradient(h1, nd):
    """Discrete gradient [nd.ndofs x h1.ndofs] of the order-1 spaces on a tetrahedral mesh as a scipy CSR matrix: the element
    matrix tet_gradient_matrix(1) with the element's orientation of every edge dof, one copy per (edge, vertex) -- all
    elements sharing an edge give the same row (+1 at the head, -1 at the tail of the global edge)."""
    import scipy.sparse as sp

    assert h1.p == 1 and nd.p == 1 and nd.diagonal_transform
    Gel = tet_gradient_matrix(1)  # [6, 4]
    sgn = np.where(np.asarray(nd.orients, dtype=bool), -1.0, 1.0)  # [ne, 6]
    rows, cols, vals = [], [], []
    for i in range(Gel.shape[0]):
        for j in range(Gel.shape[1]):
            if Gel[i, j] != 0.0:
                rows.append(nd.offsets[:, i].astype(np.int64))
                cols.append(h1.offsets[:, j].astype(np.int64))
                vals.append(sgn[:, i] * Gel[i, j])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    _, first = np.unique(rows * h1.ndofs + cols, return_index=True)
    G = sp.csr_matrix((vals[first], (rows[first], cols[first])), shape=(nd.ndofs, h1.ndofs))
    assert (np.diff(G.indptr) == 2).all() and abs(G.sum(axis=1)).max() == 0
    return G


def vertex_coordinates(h1):
    """Coordinates [h1.ndofs, 3] of the dofs of an order-1 H1 space on a tetrahedral mesh (the vertices, in its numbering)."""
    assert h1.p == 1
    xyz = np.zeros((h1.ndofs, 3))
    xyz[np.asarray(h1.offsets).ravel()] = h1.mesh.verts[h1.mesh.tets.ravel()]
    return xyz


def to_quadratic(mesh: TetMesh, warp=None) -> TetMesh:
    """tet4 -> tet10 (edge midpoints added), optionally moving all nodes by a smooth map."""
    mid = 0.5 * (mesh.verts[mesh.edge_verts[:, 0]] + mesh.verts[mesh.edge_verts[:, 1]])
    nodes = np.concatenate([mesh.verts, mid])
    en = np.concatenate([mesh.tets, mesh.nv + mesh.elem_edges], axis=1)
    if warp is not None:
        nodes = warp(nodes)
    return TetMesh(nodes[: mesh.nv], mesh.tets, mesh.attr, elem_nodes=en, nodes=nodes)


class NDTetBoundaryBlock:
    """The boundary-element block of an NDTetSpace over a set of boundary faces (what Palace builds for
    surface integrators: 2-D Nedelec triangles in 3-D space whose dofs are the adjacent tetrahedron's face /
    edge dofs, fem/libceed/restriction.cpp:15-111).  Every triangle takes its vertices in the GLOBAL
    (sorted) face frame, so its interior dof pairs coincide with the space's face dofs and only the third
    edge (G2 -> G0) is flipped: plain oriented restriction."""

    def __init__(self, space: NDTetSpace, faces, attr=None):
        from . import tri

        mesh, p = space.mesh, space.p
        self.space, self.faces = space, np.asarray(faces, dtype=np.int64)
        fv = mesh.face_verts[self.faces]                       # [nf, 3] ascending global vertex ids
        self.ne = fv.shape[0]
        self.elem = tri.NDTriElement(p)
        self.P = self.elem.P
        n_e, n_f = p, p * (p - 1)
        ekey = {tuple(e): i for i, e in enumerate(map(tuple, mesh.edge_verts))}
        off = np.zeros((self.ne, self.P), dtype=np.int64)
        ori = np.zeros((self.ne, self.P), dtype=bool)
        for k, (a, b) in enumerate(tri.LOCAL_EDGES):
            flip = a > b  # local vertices are in ascending global order
            ge = np.array([ekey[(min(f[a], f[b]), max(f[a], f[b]))] for f in fv])
            for i in range(p):
                off[:, k * p + i] = ge * n_e + (p - 1 - i if flip else i)
                ori[:, k * p + i] = flip
        for i in range(n_f):
            off[:, 3 * p + i] = space.face_base + self.faces * n_f + i
        self.offsets, self.orients = off.astype(np.int32), ori
        self.attr = np.ones(self.ne, dtype=np.int32) if attr is None else np.asarray(attr, dtype=np.int32)
        # geometry nodes: the three vertices (+ the three edge midpoints of a tet10 mesh, LOCAL_EDGES order of tri)
        if mesh.mesh_order == 1:
            self.nodes, self.elem_nodes = mesh.verts, fv
        else:
            mid = np.full(mesh.edge_verts.shape[0], -1, dtype=np.int64)
            mid[mesh.elem_edges.ravel()] = mesh.elem_nodes[:, 4:].ravel()
            em = np.stack([[mid[ekey[(min(f[a], f[b]), max(f[a], f[b]))]] for (a, b) in tri.LOCAL_EDGES] for f in fv])
            # vertex ids of the tet mesh index `verts`; tet10 geometry nodes index `nodes`: map vertices through the
            # corner entries of elem_nodes
            v2n = np.full(mesh.nv, -1, dtype=np.int64)
            v2n[mesh.tets.ravel()] = mesh.elem_nodes[:, :4].ravel()
            self.nodes, self.elem_nodes = mesh.nodes, np.concatenate([v2n[fv], em], axis=1)

    def geometry_grad_table(self, x):
        from . import tri

        return tri.TriMesh.geometry_grad_table(self, x)

    @property
    def mesh_order(self):
        return self.space.mesh.mesh_order

    def jacobians(self, x):
        """J[e, q, i, d]: 3 x 2."""
        return np.einsum("dqn,eni->eqid", self.geometry_grad_table(x), self.nodes[self.elem_nodes])


class H1TetBoundaryBlock:
    """The boundary-element block of an H1TetSpace over a set of faces (surface DiffusionIntegrator of the auxiliary-space
    operators, spaceoperator.cpp:318-330 AddAuxIntegrators): nodal H1 triangles in 3-D space whose dofs are the tetrahedral
    space's vertex / edge / face dofs.  Triangle vertices in ascending global order (the frame the space numbers its face nodes
    in), plain restriction."""

    def __init__(self, space: H1TetSpace, faces, attr=None):
        from . import tri

        mesh, p = space.mesh, space.p
        self.space, self.faces = space, np.asarray(faces, dtype=np.int64)
        fv = mesh.face_verts[self.faces]
        self.ne = fv.shape[0]
        self.elem = tri.H1TriElement(p)
        self.P = self.elem.P
        n_e, n_f = p - 1, (p - 1) * (p - 2) // 2
        ekey = {tuple(e): i for i, e in enumerate(map(tuple, mesh.edge_verts))}
        off = np.zeros((self.ne, self.P), dtype=np.int64)
        off[:, :3] = fv
        for k, (a, b) in enumerate(tri.LOCAL_EDGES):
            flip = a > b  # local vertices are in ascending global order: the space counts edge nodes from the smaller vertex
            ge = np.array([ekey[(min(f[a], f[b]), max(f[a], f[b]))] for f in fv]) if self.ne else np.zeros(0, dtype=np.int64)
            for i in range(n_e):
                off[:, 3 + k * n_e + i] = space.edge_base + ge * n_e + (n_e - 1 - i if flip else i)
        if n_f:
            fb = _interior_lattice(2, p, p - 3)
            key = {tuple(np.round(np.array(l) * p).astype(int)): m for m, l in enumerate(fb)}
            m = 0
            for jj in range(1, p):          # H1TriElement's interior nodes: (i / p, j / p), barycentric (p - i - j, i, j) / p
                for ii in range(1, p - jj):
                    off[:, 3 + 3 * n_e + m] = space.face_base + self.faces * n_f + key[(p - ii - jj, ii, jj)]
                    m += 1
        self.offsets = off.astype(np.int32)
        self.attr = np.ones(self.ne, dtype=np.int32) if attr is None else np.asarray(attr, dtype=np.int32)
        assert mesh.mesh_order == 1, "H1TetBoundaryBlock: straight-sided meshes"
        self.nodes, self.elem_nodes = mesh.verts, fv

    def geometry_grad_table(self, x):
        from . import tri

        return tri.TriMesh.geometry_grad_table(self, x)

    @property
    def mesh_order(self):
        return 1

    def jacobians(self, x):
        return np.einsum("dqn,eni->eqid", self.geometry_grad_table(x), self.nodes[self.elem_nodes])
