"""Triangular meshes and 2-D Nedelec spaces (host-side set-up for the reference's cavity2d case; the
role MFEM plays for Palace).  Order-p first-kind Nedelec triangles (p per edge, p(p-1) interior), nodal
basis dual to tangential point functionals; tri3 / tri6 geometry; conical (collapsed Gauss-Jacobi)
quadrature with p+1 points per direction (degree 2p+1 >= the default order 2p, fem/integrator.cpp:14-39).
In 2-D there is no dof transformation (fem/libceed/restriction.cpp:234-238): the restriction is the
oriented one."""
from __future__ import annotations

import struct

import numpy as np

from .basis1d import gauss_legendre

REF_VERTS = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
LOCAL_EDGES = [(0, 1), (1, 2), (2, 0)]
_C = np.array([1.0 / 3.0, 1.0 / 3.0])


def tri_quadrature(n):
    from scipy.special import roots_jacobi

    t1, w1 = roots_jacobi(n, 1.0, 0.0)
    t0, w0 = roots_jacobi(n, 0.0, 0.0)
    u, wu = 0.5 * (1 + t1), w1 / 4.0
    v, wv = 0.5 * (1 + t0), w0 / 2.0
    pts = np.array([[u[a], v[b] * (1 - u[a])] for a in range(n) for b in range(n)])
    wts = np.array([wu[a] * wv[b] for a in range(n) for b in range(n)])
    return pts, wts


def _monos(lo, hi):
    return [(a, d - a) for d in range(lo, hi + 1) for a in range(d + 1)]


def _mv(e, z):
    return z[..., 0] ** e[0] * z[..., 1] ** e[1]


def _mg(e, z):
    g = np.zeros(z.shape)
    if e[0] > 0:
        g[..., 0] = e[0] * _mv((e[0] - 1, e[1]), z)
    if e[1] > 0:
        g[..., 1] = e[1] * _mv((e[0], e[1] - 1), z)
    return g


class NDTriElement:
    """R_p = P_{p-1}^2 + (-z_y, z_x) P~_{p-1} on the reference triangle (z = centred coordinates)."""

    def __init__(self, p):
        self.p, self.P = p, p * (p + 2)
        # candidates as lists of (component, exponent, coefficient)
        self.cands = [[(c, e, 1.0)] for e in _monos(0, p - 1) for c in range(2)]
        for e in _monos(p - 1, p - 1):
            self.cands.append([(0, (e[0], e[1] + 1), -1.0), (1, (e[0] + 1, e[1]), 1.0)])
        eo = gauss_legendre(p)[0]
        pts, tans = [], []
        for a, b in LOCAL_EDGES:
            for t in eo:
                pts.append((1 - t) * REF_VERTS[a] + t * REF_VERTS[b])
                tans.append(REF_VERTS[b] - REF_VERTS[a])
        if p >= 2:
            for i in range(p - 1):
                for j in range(p - 1 - i):
                    k = p - 2 - i - j
                    l = np.array([i + 1, j + 1, k + 1]) / (p + 1)
                    x = l[0] * REF_VERTS[0] + l[1] * REF_VERTS[1] + l[2] * REF_VERTS[2]
                    pts += [x, x]
                    tans += [np.array([1.0, 0.0]), np.array([0.0, 1.0])]
        self.dof_pts, self.dof_tans = np.array(pts), np.array(tans)
        assert len(pts) == self.P
        Vm = np.array([np.einsum("nd,nd->n", self._eval(c, self.dof_pts), self.dof_tans) for c in self.cands]).T
        self.coef = np.linalg.pinv(Vm, rcond=1e-12)
        assert np.abs(Vm @ self.coef - np.eye(self.P)).max() < 1e-10

    @staticmethod
    def _eval(c, x):
        z = x - _C
        v = np.zeros(z.shape)
        for comp, e, a in c:
            v[..., comp] += a * _mv(e, z)
        return v

    @staticmethod
    def _curl(c, x):
        z = x - _C
        w = np.zeros(z.shape[:-1])
        for comp, e, a in c:
            g = a * _mg(e, z)
            w += g[..., 0] if comp == 1 else -g[..., 1]   # d_x v_y - d_y v_x
        return w

    def tables(self, x):
        """interp [2, Q, P], curl [1, Q, P]."""
        val = np.array([self._eval(c, x) for c in self.cands])       # [k, Q, 2]
        cur = np.array([self._curl(c, x) for c in self.cands])       # [k, Q]
        return (np.ascontiguousarray(np.einsum("kqd,kj->dqj", val, self.coef)),
                np.ascontiguousarray(np.einsum("kq,kj->qj", cur, self.coef)[None]))


class TriMesh:
    """tris [ne, 3] vertex ids; geometry nodes order 1 (vertices) or 2 (tri6: + edge midpoints in
    LOCAL_EDGES order)."""

    def __init__(self, verts, tris, attr=None, elem_nodes=None, nodes=None, bdr_edges=None, bdr_attr=None):
        self.verts, self.tris = np.asarray(verts, dtype=np.float64), np.asarray(tris, dtype=np.int64)
        self.ne = self.tris.shape[0]
        self.attr = np.ones(self.ne, dtype=np.int32) if attr is None else np.asarray(attr, dtype=np.int32)
        self.nodes = self.verts if nodes is None else np.asarray(nodes, dtype=np.float64)
        self.elem_nodes = self.tris if elem_nodes is None else np.asarray(elem_nodes, dtype=np.int64)
        self.mesh_order = 1 if self.elem_nodes.shape[1] == 3 else 2
        self.bdr_edges, self.bdr_attr = bdr_edges, bdr_attr
        X = self.verts[self.tris]
        d1, d2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
        if np.any(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0] <= 0):
            raise ValueError("inverted triangles")
        e = np.stack([np.sort(self.tris[:, list(le)], axis=1) for le in LOCAL_EDGES], axis=1)
        ue, inv, cnt = np.unique(e.reshape(-1, 2), axis=0, return_inverse=True, return_counts=True)
        self.edge_verts, self.elem_edges = ue, inv.reshape(self.ne, 3)
        self.boundary_edge_mask = cnt == 1

    def geometry_grad_table(self, x):
        l = np.stack([1 - x.sum(axis=1), x[:, 0], x[:, 1]], axis=1)
        dl = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
        Q = x.shape[0]
        if self.mesh_order == 1:
            return np.ascontiguousarray(np.broadcast_to(dl.T[:, None, :], (2, Q, 3)))
        G = np.zeros((2, Q, 6))
        for i in range(3):
            G[:, :, i] = ((4 * l[:, i] - 1)[None, :]) * dl[i][:, None]
        for k, (a, b) in enumerate(LOCAL_EDGES):
            G[:, :, 3 + k] = 4 * (l[:, a][None, :] * dl[b][:, None] + l[:, b][None, :] * dl[a][:, None])
        return G

    def jacobians(self, x):
        return np.einsum("dqn,eni->eqid", self.geometry_grad_table(x), self.nodes[self.elem_nodes])


def read_gmsh22_tris(path):
    """Gmsh 2.2 (binary / ASCII) with tri3 (type 2) or tri6 (type 9) elements and 2- / 3-node boundary
    lines (types 1 / 8): the reference's examples/cavity2d/mesh/cavity2d.msh."""
    data = open(path, "rb").read()

    def section(name):
        a = data.index(b"$" + name + b"\n") + len(name) + 2
        return a, data.index(b"$End" + name)

    a, _ = section(b"MeshFormat")
    binary = int(data[a : data.index(b"\n", a)].split()[1]) == 1
    a, b = section(b"Nodes")
    nl = data.index(b"\n", a)
    nn = int(data[a:nl])
    if binary:
        rec = np.frombuffer(data, dtype=np.dtype([("i", "<i4"), ("x", "<f8", 3)]), count=nn, offset=nl + 1)
        ids, xyz = rec["i"].astype(np.int64), rec["x"].copy()
    else:
        rows = np.array(data[nl + 1 : b].split(), dtype=np.float64).reshape(nn, 4)
        ids, xyz = rows[:, 0].astype(np.int64), rows[:, 1:]
    idmap = np.full(ids.max() + 1, -1, dtype=np.int64)
    idmap[ids] = np.arange(nn)
    a, b = section(b"Elements")
    nl = data.index(b"\n", a)
    nelem = int(data[a:nl])
    npe = {2: 3, 9: 6, 1: 2, 8: 3, 15: 1}
    tri, tattr, lin, lattr = [], [], [], []
    if binary:
        off, done = nl + 1, 0
        while done < nelem:
            et, nf, nt = struct.unpack_from("<iii", data, off)
            off += 12
            w = 1 + nt + npe[et]
            rec = np.frombuffer(data, dtype="<i4", count=w * nf, offset=off).reshape(nf, w)
            off += 4 * w * nf
            if et in (2, 9):
                tri.append(rec[:, 1 + nt :]), tattr.append(rec[:, 1])
            elif et in (1, 8):
                lin.append(rec[:, 1 + nt :]), lattr.append(rec[:, 1])
            done += nf
    else:
        for line in data[nl + 1 : b].splitlines():
            r = [int(v) for v in line.split()]
            if not r:
                continue
            et, nt = r[1], r[2]
            if et in (2, 9):
                tri.append(np.array([r[3 + nt :]])), tattr.append(np.array([r[3]]))
            elif et in (1, 8):
                lin.append(np.array([r[3 + nt :]])), lattr.append(np.array([r[3]]))
    en = idmap[np.concatenate(tri)]
    X = xyz[en[:, :3], :2]
    d1, d2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    neg = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0] < 0
    if np.any(neg):  # swap vertices 1 <-> 2 (gmsh tri6 edges (0,1),(1,2),(2,0) -> (0,2),(2,1),(1,0))
        sw = [0, 2, 1] if en.shape[1] == 3 else [0, 2, 1, 5, 4, 3]
        en[neg] = en[neg][:, sw]
    t = en[:, :3]
    used, inv = np.unique(t, return_inverse=True)
    vmap = np.full(nn, -1, dtype=np.int64)
    vmap[used] = np.arange(used.size)
    be = vmap[idmap[np.concatenate(lin)][:, :2]] if lin else None
    quad = en.shape[1] == 6
    return TriMesh(xyz[used, :2], inv.reshape(-1, 3), np.concatenate(tattr), elem_nodes=en if quad else None,
                   nodes=xyz[:, :2] if quad else None, bdr_edges=be, bdr_attr=np.concatenate(lattr) if lin else None)


class NDTriSpace:
    """Order-p Nedelec space on a TriMesh: global dofs = edges | interiors; oriented restriction."""

    def __init__(self, mesh: TriMesh, p: int):
        self.mesh, self.p = mesh, p
        self.elem = NDTriElement(p)
        self.P = self.elem.P
        ne, t = mesh.ne, mesh.tris
        n_i = p * (p - 1)
        NE_ = mesh.edge_verts.shape[0]
        self.ndofs = NE_ * p + ne * n_i
        off = np.zeros((ne, self.P), dtype=np.int64)
        ori = np.zeros((ne, self.P), dtype=bool)
        for k, (a, b) in enumerate(LOCAL_EDGES):
            flip = t[:, a] > t[:, b]
            for i in range(p):
                off[:, k * p + i] = mesh.elem_edges[:, k] * p + np.where(flip, p - 1 - i, i)
                ori[:, k * p + i] = flip
        for i in range(n_i):
            off[:, 3 * p + i] = NE_ * p + np.arange(ne) * n_i + i
        self.offsets, self.orients = off.astype(np.int32), ori

    def ess_dofs(self, edge_mask=None):
        m = self.mesh
        em = m.boundary_edge_mask if edge_mask is None else edge_mask
        edges = np.nonzero(em)[0]
        return (edges[:, None] * self.p + np.arange(self.p)[None, :]).ravel().astype(np.int32)


class H1TriElement:
    """Nodal H1 triangle of order p on the lattice nodes: vertices, p - 1 per edge (LOCAL_EDGES order, from the edge's
    first to its second local vertex), interior.  tables(x) -> interp [1, Q, P], grad [2, Q, P]."""

    def __init__(self, p):
        self.p = p
        V = REF_VERTS
        nodes = [V[0], V[1], V[2]]
        for a, b in LOCAL_EDGES:
            for k in range(1, p):
                nodes.append(V[a] + (V[b] - V[a]) * k / p)
        for j in range(1, p):
            for i in range(1, p - j):
                nodes.append(np.array([i / p, j / p]))
        self.nodes = np.array(nodes)
        self.P = self.nodes.shape[0]
        assert self.P == (p + 1) * (p + 2) // 2
        self.monos = _monos(0, p)
        Vm = np.array([_mv(e, self.nodes - _C) for e in self.monos]).T
        self.coef = np.linalg.inv(Vm)

    def tables(self, x):
        z = x - _C
        val = np.array([_mv(e, z) for e in self.monos])       # [nm, Q]
        grd = np.array([_mg(e, z) for e in self.monos])       # [nm, Q, 2]
        interp = np.einsum("kq,kj->qj", val, self.coef)[None]
        grad = np.einsum("kqd,kj->dqj", grd, self.coef)
        return np.ascontiguousarray(interp), np.ascontiguousarray(grad)


class H1TriSpace:
    """Order-p nodal H1 space on a TriMesh: global dofs = vertices | edges | interiors, plain restriction."""

    def __init__(self, mesh: TriMesh, p: int):
        self.mesh, self.p = mesh, p
        self.elem = H1TriElement(p)
        self.P = self.elem.P
        ne, nv, ned = mesh.ne, mesh.verts.shape[0], mesh.edge_verts.shape[0]
        n_e, n_i = p - 1, (p - 1) * (p - 2) // 2
        self.edge_base, self.int_base = nv, nv + ned * n_e
        self.ndofs = self.int_base + ne * n_i
        off = np.zeros((ne, self.P), dtype=np.int64)
        t = mesh.tris
        off[:, :3] = t
        for k, (a, b) in enumerate(LOCAL_EDGES):
            flip = t[:, a] > t[:, b]
            for m in range(n_e):
                gm = np.where(flip, n_e - 1 - m, m)
                off[:, 3 + k * n_e + m] = self.edge_base + mesh.elem_edges[:, k] * n_e + gm
        for i in range(n_i):
            off[:, 3 + 3 * n_e + i] = self.int_base + np.arange(ne) * n_i + i
        self.offsets = off.astype(np.int32)

    def ess_dofs(self, edge_mask=None):
        m = self.mesh
        em = m.boundary_edge_mask if edge_mask is None else edge_mask
        edges = np.nonzero(em)[0]
        n_e = self.p - 1
        d = [np.unique(m.edge_verts[edges].ravel()),
             (self.edge_base + edges[:, None] * n_e + np.arange(n_e)[None, :]).ravel()]
        return np.unique(np.concatenate(d)).astype(np.int32)



def nd_tri_transfer_matrix(pc, pf):
    """Element matrix of the p-prolongation ND(pc) -> ND(pf) on the triangle (MFEM GetTransferMatrix, basis.cpp:132-138):
    the fine dof functionals applied to the coarse basis, [P_f, P_c]."""
    fine = NDTriElement(pf)
    interp, _ = NDTriElement(pc).tables(fine.dof_pts)      # [2, P_f, P_c]
    return np.ascontiguousarray(np.einsum("dji,jd->ji", interp, fine.dof_tans))


def tri_gradient_matrix(p):
    """Element matrix of the discrete gradient H1(p) -> ND(p) on the triangle (basis.cpp:139-143), [P_nd, P_h1]."""
    nd = NDTriElement(p)
    _, grad = H1TriElement(p).tables(nd.dof_pts)           # [2, P_nd, P_h1]
    return np.ascontiguousarray(np.einsum("dji,jd->ji", grad, nd.dof_tans))


def restriction(space):
    """dict(offsets, lsize[, orients]) of a triangle space as the C ABI's dense interpolators take it (the 2-D restriction is
    the oriented one, its own inverse: fem/libceed/restriction.cpp:234-238)."""
    r = dict(offsets=space.offsets, lsize=space.ndofs)
    if getattr(space, "orients", None) is not None:
        r["orients"] = space.orients
    return r


def lowest_order_gradient(h1, nd):
    """Discrete gradient [nd.ndofs x h1.ndofs] of the order-1 spaces on a TriMesh as a scipy CSR matrix (+1 at the head, -1 at
    the tail of every global edge), built from the element matrix and the element's edge orientations like the tetrahedral one
    (tet.lowest_order_gradient)."""
    import scipy.sparse as sp

    assert h1.p == 1 and nd.p == 1
    Gel = tri_gradient_matrix(1)  # [3, 3]
    sgn = np.where(np.asarray(nd.orients, dtype=bool), -1.0, 1.0)
    rows, cols, vals = [], [], []
    for i in range(Gel.shape[0]):
        for j in range(Gel.shape[1]):
            if abs(Gel[i, j]) > 1e-14:
                rows.append(nd.offsets[:, i].astype(np.int64))
                cols.append(h1.offsets[:, j].astype(np.int64))
                vals.append(sgn[:, i] * Gel[i, j])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    _, first = np.unique(rows * h1.ndofs + cols, return_index=True)
    G = sp.csr_matrix((vals[first], (rows[first], cols[first])), shape=(nd.ndofs, h1.ndofs))
    assert (np.diff(G.indptr) == 2).all() and abs(G.sum(axis=1)).max() < 1e-13
    return G


def vertex_coordinates(h1):
    """Coordinates [h1.ndofs, 2] of the dofs of an order-1 H1 space on a TriMesh."""
    assert h1.p == 1
    xy = np.zeros((h1.ndofs, 2))
    xy[np.asarray(h1.offsets).ravel()] = h1.mesh.verts[h1.mesh.tris.ravel()]
    return xy


def edge_current_load(nd, bdr_verts, bdr_attr, source_attr, direction):
    """Load vector of a unit surface current on the boundary edges with attribute `source_attr` (the reference's
    SurfaceCurrentOperator on a 2-D mesh, models/surfacecurrentoperator.cpp: b_i = int_Gamma J_s . phi_i ds with J_s the
    unit vector `direction` along the edge) and the mask of the remaining boundary edges.  bdr_verts [nb, 2]: vertex ids
    (the space's mesh numbering) of every boundary edge.  Returns (b, other_edge_mask, total source length)."""
    mesh = nd.mesh
    ekey = {tuple(e): i for i, e in enumerate(map(tuple, mesh.edge_verts))}
    owner = {}
    for e in range(mesh.ne):
        for k in range(3):
            owner.setdefault(int(mesh.elem_edges[e, k]), (e, k))
    s, ws = gauss_legendre(nd.p + 2)
    sgn = np.where(nd.orients, -1.0, 1.0)
    b = np.zeros(nd.ndofs)
    other = np.zeros(mesh.edge_verts.shape[0], dtype=bool)
    length = 0.0
    direction = np.asarray(direction, dtype=np.float64)
    for (va, vb), attr in zip(bdr_verts, bdr_attr):
        ge = ekey[(min(va, vb), max(va, vb))]
        if attr != source_attr:
            other[ge] = True
            continue
        e, k = owner[ge]
        la, lb = LOCAL_EDGES[k]
        that = REF_VERTS[lb] - REF_VERTS[la]
        xs = REF_VERTS[la][None, :] + s[:, None] * that[None, :]
        val, _ = nd.elem.tables(xs)                       # [2, nq, P]
        loc = np.einsum("dqj,d,q->j", val, that, ws)      # int phi_hat . t_hat ds_hat (= int phi . t ds: covariant Piola)
        tphys = mesh.verts[mesh.tris[e, lb]] - mesh.verts[mesh.tris[e, la]]
        length += float(np.linalg.norm(tphys))
        np.add.at(b, nd.offsets[e], np.sign(tphys @ direction) * sgn[e] * loc)
    return b, other, length
