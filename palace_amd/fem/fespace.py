"""Nedelec (H(curl)) and H1 spaces of order p on hex meshes: element->dof maps with orientation.

Stand-in for what Palace obtains from `mfem::FiniteElementSpace::GetElementDofs` and turns into
libCEED element restrictions (reference: palace/fem/libceed/restriction.cpp:207-385 — index
array `tp_el_dof[j + P*e]` plus, for hexes, a `bool` orientation per local dof, :288-298,:370-377;
H1 tensor elements use the lexicographic variant :113-205).  MFEM is not in the reference tree, so
the numbering below is defined here; it has the same structure (edge dofs, then face dofs, then
interior dofs; a flipped dof has sign -1), and every basis-dependent quantity is pinned only
through basis-invariant results (eigenfrequencies, K*grad = 0, symmetry).

Local tensor ("lexicographic") ordering of the Nedelec hex element of order p, P = 3 p (p+1)^2:
  x-block: ob_i(x) cb_j(y) cb_k(z) e_x, index i + p (j + (p+1) k),       i<p,  j,k<=p
  y-block: cb_i(x) ob_j(y) cb_k(z) e_y, index i + (p+1) (j + p k)
  z-block: cb_i(x) cb_j(y) ob_k(z) e_z, index i + (p+1) (j + (p+1) k)
with cb = closed Gauss-Lobatto and ob = open Gauss-Legendre nodal bases.
"""
from __future__ import annotations

import numpy as np

from .mesh import HEX_EDGES, HEX_FACE_AXES, HEX_FACES_UV, HexMesh

_VERT_IJK = np.array(
    [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]
)


def nd_block_shape(p: int, comp: int):
    """(nx, ny, nz) of the component block `comp` of the ND hex element."""
    n = [p + 1, p + 1, p + 1]
    n[comp] = p
    return tuple(n)


def nd_lex_index(p: int, comp: int, i, j, k):
    nx, ny, _ = nd_block_shape(p, comp)
    return comp * p * (p + 1) ** 2 + i + nx * (j + ny * k)


def _face_orientation(fv: np.ndarray):
    """For face corner ids fv[..., 4] = (c00, c10, c01, c11) return (ou, ov, swap): the global
    face frame has its origin at the corner with the smallest id, s-axis towards the adjacent
    corner with the smaller id."""
    org = np.argmin(fv, axis=-1)
    ou = org & 1
    ov = org >> 1
    # neighbours of the origin along u and along v
    nb_u = np.take_along_axis(fv, ((1 - ou) + 2 * ov)[..., None], axis=-1)[..., 0]
    nb_v = np.take_along_axis(fv, (ou + 2 * (1 - ov))[..., None], axis=-1)[..., 0]
    swap = nb_v < nb_u
    return ou.astype(bool), ov.astype(bool), swap


class NDHexSpace:
    """Order-p Nedelec space on a HexMesh.

    elem_dof_lex  [NE, P] int32  global dof per local tensor dof
    elem_sign_lex [NE, P] int8   +1 / -1
    ndofs, and ess_dofs(): dofs on boundary faces (PEC).
    """

    def __init__(self, mesh: HexMesh, p: int):
        self.mesh, self.p = mesh, p
        self.P = 3 * p * (p + 1) ** 2
        ne = mesh.ne
        n_e, n_f, n_i = p, 2 * p * (p - 1), 3 * p * (p - 1) ** 2
        self.n_per = (n_e, n_f, n_i)
        self.edge_base = 0
        self.face_base = mesh.nedges * n_e
        self.int_base = self.face_base + mesh.nfaces * n_f
        self.ndofs = self.int_base + ne * n_i
        if self.ndofs >= 2**31:
            raise ValueError("dof count exceeds int32")
        dof = np.full((ne, self.P), -1, dtype=np.int64)
        sgn = np.ones((ne, self.P), dtype=np.int8)
        verts = mesh.verts

        # --- edges
        m = np.arange(p)
        for le, (a, b) in enumerate(HEX_EDGES):
            d = _VERT_IJK[b] - _VERT_IJK[a]
            comp = int(np.argmax(d))
            ijk = [None, None, None]
            for ax in range(3):
                ijk[ax] = m if ax == comp else np.full(p, _VERT_IJK[a][ax] * p)
            lex = nd_lex_index(p, comp, ijk[0], ijk[1], ijk[2])
            flip = verts[:, a] > verts[:, b]
            gm = np.where(flip[:, None], p - 1 - m[None, :], m[None, :])
            dof[:, lex] = self.edge_base + mesh.elem_edges[:, le, None] * n_e + gm
            sgn[:, lex] = np.where(flip[:, None], -1, 1)

        # --- faces
        if p > 1:
            for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
                fv = verts[:, HEX_FACES_UV[lf]]
                ou, ov, swap = _face_orientation(fv)
                base = self.face_base + mesh.elem_faces[:, lf] * n_f
                ou_, ov_, sw_ = ou[:, None], ov[:, None], swap[:, None]
                # u-directed: open index a along u, interior closed index b along v
                A, B = np.meshgrid(np.arange(p), np.arange(1, p), indexing="ij")
                A, B = A.ravel(), B.ravel()
                ijk = [None, None, None]
                ijk[nax] = np.full(A.size, side * p)
                ijk[uax], ijk[vax] = A, B
                lex = nd_lex_index(p, uax, ijk[0], ijk[1], ijk[2])
                a2 = np.where(ou_, p - 1 - A[None, :], A[None, :])
                b2 = np.where(ov_, p - B[None, :], B[None, :])
                g_s = a2 + p * (b2 - 1)
                g_t = p * (p - 1) + (b2 - 1) + (p - 1) * a2
                dof[:, lex] = base[:, None] + np.where(sw_, g_t, g_s)
                sgn[:, lex] = np.where(ou_, -1, 1)
                # v-directed: interior closed index a along u, open index b along v
                A, B = np.meshgrid(np.arange(1, p), np.arange(p), indexing="ij")
                A, B = A.ravel(), B.ravel()
                ijk[uax], ijk[vax] = A, B
                lex = nd_lex_index(p, vax, ijk[0], ijk[1], ijk[2])
                a2 = np.where(ou_, p - A[None, :], A[None, :])
                b2 = np.where(ov_, p - 1 - B[None, :], B[None, :])
                g_t = p * (p - 1) + (a2 - 1) + (p - 1) * b2
                g_s = b2 + p * (a2 - 1)
                dof[:, lex] = base[:, None] + np.where(sw_, g_s, g_t)
                sgn[:, lex] = np.where(ov_, -1, 1)

            # --- interior
            cnt = 0
            for comp in range(3):
                rng = [np.arange(1, p)] * 3
                rng[comp] = np.arange(p)
                K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
                lex = nd_lex_index(p, comp, I.ravel(), J.ravel(), K.ravel())
                loc = cnt + np.arange(lex.size)
                dof[:, lex] = self.int_base + np.arange(ne)[:, None] * n_i + loc[None, :]
                cnt += lex.size
        assert dof.min() >= 0
        self.elem_dof_lex = dof.astype(np.int32)
        self.elem_sign_lex = sgn

    def ess_dofs(self, face_mask=None) -> np.ndarray:
        """Sorted global dofs with vanishing tangential trace on the mesh boundary (PEC).
        face_mask [nfaces] overrides the set of boundary faces (default: faces with one element)."""
        mesh, p = self.mesh, self.p
        fm = mesh.boundary_face_mask if face_mask is None else face_mask
        bmask = fm[mesh.elem_faces]  # [NE, 6]
        out = []
        for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
            el = np.nonzero(bmask[:, lf])[0]
            if el.size == 0:
                continue
            for comp, oax in ((uax, vax), (vax, uax)):
                rng = [None, None, None]
                rng[nax] = np.array([side * p])
                rng[comp] = np.arange(p)
                rng[oax] = np.arange(p + 1)
                K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
                lex = nd_lex_index(p, comp, I.ravel(), J.ravel(), K.ravel())
                out.append(self.elem_dof_lex[el][:, lex].ravel())
        return np.unique(np.concatenate(out)).astype(np.int32) if out else np.zeros(0, np.int32)

    def dof_map_native(self):
        """lex -> native signed map in the sense of MFEM's `TensorBasisElement::GetDofMap()`
        (entry -1-n means native dof n with flipped sign).  Native order: 12 edges x p, 6 faces x
        2p(p-1) (first-tangent dofs then second-tangent dofs in the face's own frame defined by its
        MFEM vertex tuple), then the interior by component.  This is what the dense tables and the
        native-ordered restriction handed through the C boundary are expressed in."""
        p = self.p
        nat = np.zeros(self.P, dtype=np.int64)
        o = 0
        m = np.arange(p)
        for a, b in HEX_EDGES:
            d = _VERT_IJK[b] - _VERT_IJK[a]
            comp = int(np.argmax(d))
            ijk = [m if ax == comp else np.full(p, _VERT_IJK[a][ax] * p) for ax in range(3)]
            lex = nd_lex_index(p, comp, *ijk)
            nat[lex] = o + m
            o += p
        # MFEM face vertex tuples; frame: first axis v0->v1, second axis v0->v3
        mfem_faces = [(3, 2, 1, 0), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)]
        for f in mfem_faces:
            v0 = _VERT_IJK[f[0]]
            d1 = _VERT_IJK[f[1]] - v0
            d2 = _VERT_IJK[f[3]] - v0
            ax1, ax2 = int(np.argmax(np.abs(d1))), int(np.argmax(np.abs(d2)))
            s1, s2 = int(d1[ax1]), int(d2[ax2])
            nax = 3 - ax1 - ax2
            # first-tangent dofs: open along ax1 (a), closed interior along ax2 (b)
            for (oax, cax, so, sc) in ((ax1, ax2, s1, s2), (ax2, ax1, s2, s1)):
                for b in range(1, p):
                    for a in range(p):
                        ijk = [0, 0, 0]
                        ijk[nax] = v0[nax] * p
                        ijk[oax] = a if so > 0 else p - 1 - a
                        ijk[cax] = b if sc > 0 else p - b
                        lex = nd_lex_index(p, oax, *ijk)
                        nat[lex] = o if so > 0 else -1 - o
                        o += 1
        for comp in range(3):
            rng = [np.arange(1, p)] * 3
            rng[comp] = np.arange(p)
            K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
            lex = nd_lex_index(p, comp, I.ravel(), J.ravel(), K.ravel())
            nat[lex] = o + np.arange(lex.size)
            o += lex.size
        assert o == self.P
        return nat

    def native_restriction(self):
        """(offsets [NE,P] int32, orients [NE,P] bool) in native local order — exactly the two
        arrays Palace passes to `CeedElemRestrictionCreateOriented` (restriction.cpp:370-377)."""
        nat = self.dof_map_native()
        idx = np.where(nat >= 0, nat, -1 - nat)
        flip = nat < 0
        off = np.empty_like(self.elem_dof_lex)
        ori = np.empty(self.elem_dof_lex.shape, dtype=bool)
        off[:, idx] = self.elem_dof_lex
        ori[:, idx] = (self.elem_sign_lex < 0) ^ flip[None, :]
        return off, ori


class H1HexSpace:
    """Order-p H1 space (Gauss-Lobatto nodal, tensor, lexicographic local order; Palace uses the
    lexicographic restriction for scalar tensor elements, restriction.cpp:113-205).

    elem_dof_lex [NE, (p+1)^3] int32; no signs.
    """

    def __init__(self, mesh: HexMesh, p: int):
        self.mesh, self.p = mesh, p
        n1 = p + 1
        self.P = n1**3
        ne = mesh.ne
        n_e, n_f, n_i = p - 1, (p - 1) ** 2, (p - 1) ** 3
        self.vert_base = 0
        self.edge_base = mesh.nv
        self.face_base = self.edge_base + mesh.nedges * n_e
        self.int_base = self.face_base + mesh.nfaces * n_f
        self.ndofs = self.int_base + ne * n_i
        dof = np.full((ne, self.P), -1, dtype=np.int64)
        verts = mesh.verts

        def lexi(i, j, k):
            return i + n1 * (j + n1 * k)

        for lv in range(8):
            i, j, k = _VERT_IJK[lv] * p
            dof[:, lexi(i, j, k)] = verts[:, lv]
        if p > 1:
            m = np.arange(1, p)
            for le, (a, b) in enumerate(HEX_EDGES):
                d = _VERT_IJK[b] - _VERT_IJK[a]
                comp = int(np.argmax(d))
                ijk = [m if ax == comp else np.full(p - 1, _VERT_IJK[a][ax] * p) for ax in range(3)]
                lex = lexi(*ijk)
                flip = verts[:, a] > verts[:, b]
                gm = np.where(flip[:, None], p - m[None, :], m[None, :]) - 1
                dof[:, lex] = self.edge_base + mesh.elem_edges[:, le, None] * n_e + gm
            for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
                fv = verts[:, HEX_FACES_UV[lf]]
                ou, ov, swap = _face_orientation(fv)
                A, B = np.meshgrid(m, m, indexing="ij")
                A, B = A.ravel(), B.ravel()
                ijk = [None, None, None]
                ijk[nax] = np.full(A.size, side * p)
                ijk[uax], ijk[vax] = A, B
                lex = lexi(*ijk)
                a2 = np.where(ou[:, None], p - A[None, :], A[None, :])
                b2 = np.where(ov[:, None], p - B[None, :], B[None, :])
                s = np.where(swap[:, None], b2, a2)
                t = np.where(swap[:, None], a2, b2)
                dof[:, lex] = self.face_base + mesh.elem_faces[:, lf, None] * n_f + (s - 1) + (p - 1) * (t - 1)
            K, J, I = np.meshgrid(m, m, m, indexing="ij")
            lex = lexi(I.ravel(), J.ravel(), K.ravel())
            dof[:, lex] = self.int_base + np.arange(ne)[:, None] * n_i + np.arange(lex.size)[None, :]
        assert dof.min() >= 0
        self.elem_dof_lex = dof.astype(np.int32)

    def ess_dofs(self) -> np.ndarray:
        mesh, p = self.mesh, self.p
        n1 = p + 1
        bmask = mesh.boundary_face_mask[mesh.elem_faces]
        out = []
        for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
            el = np.nonzero(bmask[:, lf])[0]
            if el.size == 0:
                continue
            rng = [np.arange(n1)] * 3
            rng[nax] = np.array([side * p])
            K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
            lex = (I + n1 * (J + n1 * K)).ravel()
            out.append(self.elem_dof_lex[el][:, lex].ravel())
        return np.unique(np.concatenate(out)).astype(np.int32) if out else np.zeros(0, np.int32)


class NDHexBoundaryBlock:
    """Boundary-element block of an NDHexSpace over a set of boundary faces: 2-D Nedelec quadrilaterals in 3-D
    space whose dofs are the adjacent hexahedron's dofs on that face (what Palace builds for surface integrators,
    fem/libceed/restriction.cpp:15-111 `GetFaceDofsFromAdjacentElement`).  The quad inherits the (u, v) axes of
    the element's local face, so its tensor dofs are the element's lexicographic face dofs, signs included.

    Local dof order: u-component block a + p b (a open along u, b closed along v), then the v-component block
    a + (p + 1) b (a closed along u, b open along v).  Quadrature: q1d x q1d tensor rule, q = qu + q1d qv."""

    def __init__(self, space: NDHexSpace, face_mask=None, attr=None):
        from .mesh import HEX_FACE_AXES

        mesh, p = space.mesh, space.p
        self.space, self.p = space, p
        fm = mesh.boundary_face_mask if face_mask is None else face_mask
        bmask = fm[mesh.elem_faces]
        offs, signs, nodes, faces = [], [], [], []
        for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
            el = np.nonzero(bmask[:, lf])[0]
            if el.size == 0:
                continue
            lex = []
            for comp, na, nb in ((uax, p, p + 1), (vax, p + 1, p)):
                B, A = np.meshgrid(np.arange(nb), np.arange(na), indexing="ij")   # a fastest
                ijk = [None, None, None]
                ijk[nax] = np.full(A.size, side * p)
                ijk[uax], ijk[vax] = A.ravel(), B.ravel()
                lex.append(nd_lex_index(p, comp, *ijk))
            lex = np.concatenate(lex)
            offs.append(space.elem_dof_lex[el][:, lex])
            signs.append(space.elem_sign_lex[el][:, lex])
            iv, iu = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")   # node n = iu + 3 iv
            nijk = [None, None, None]
            nijk[nax] = np.full(9, side * 2)
            nijk[uax], nijk[vax] = iu.ravel(), iv.ravel()
            lat = nijk[0] + 3 * (nijk[1] + 3 * nijk[2])
            nodes.append(mesh.elem_nodes[el][:, lat])
            faces.append(mesh.elem_faces[el, lf])
        self.offsets = np.concatenate(offs).astype(np.int32)
        self.orients = np.concatenate(signs) < 0
        self.elem_nodes = np.concatenate(nodes).astype(np.int64)
        self.faces = np.concatenate(faces)
        self.nodes = mesh.x
        self.ne, self.P = self.offsets.shape
        self.attr = np.ones(self.ne, dtype=np.int32) if attr is None else np.asarray(attr, dtype=np.int32)

    def tables(self, q1d):
        """(interp [2, Q, P], mesh_grad [2, Q, 9], weights [Q]) of the quad at the q1d x q1d Gauss-Legendre rule."""
        from .basis1d import Tables1D
        from .mesh import _q2_1d

        p = self.p
        t = Tables1D(p, q1d)
        Q = q1d * q1d
        interp = np.zeros((2, Q, self.P))
        for qv in range(q1d):
            for qu in range(q1d):
                q = qu + q1d * qv
                interp[0, q, : p * (p + 1)] = np.outer(t.Bc[qv], t.Bo[qu]).ravel()     # index a + p b
                interp[1, q, p * (p + 1):] = np.outer(t.Bo[qv], t.Bc[qu]).ravel()      # index a + (p + 1) b
        B2, G2 = _q2_1d(t.qx)
        grad = np.zeros((2, Q, 9))
        for qv in range(q1d):
            for qu in range(q1d):
                q = qu + q1d * qv
                grad[0, q] = np.outer(B2[qv], G2[qu]).ravel()   # node iu + 3 iv
                grad[1, q] = np.outer(G2[qv], B2[qu]).ravel()
        w = np.outer(t.qw, t.qw).ravel()
        return interp, grad, w


# ---- lowest-order auxiliary-space data (what the reference hands to HYPRE's AMS, linalg/ams.cpp:46-100) ------------------

def lowest_order_gradient(h1, nd):
    """Discrete gradient G [nd.ndofs x h1.ndofs] of the order-1 spaces on a hex mesh as a scipy CSR matrix: the edge dof of
    the lowest-order Nedelec element is the difference of the nodal values at its end points, taken in the direction of the
    reference edge and signed with the element's orientation of the dof (every element sharing an edge gives the same row)."""
    import scipy.sparse as sp

    assert h1.p == 1 and nd.p == 1
    ne = nd.elem_dof_lex.shape[0]
    rows, cols, vals = [], [], []

    def lex(i, j, k):
        return i + 2 * (j + 2 * k)

    loc = []  # (local ND dof, local H1 tail, local H1 head), ND lexicographic order: component-major, first index fastest
    for k in range(2):
        for j in range(2):
            loc.append((len(loc), lex(0, j, k), lex(1, j, k)))  # x-directed edges, dof (j, k)
    for k in range(2):
        for i in range(2):
            loc.append((len(loc), lex(i, 0, k), lex(i, 1, k)))  # y-directed, dof (i, k)
    for j in range(2):
        for i in range(2):
            loc.append((len(loc), lex(i, j, 0), lex(i, j, 1)))  # z-directed, dof (i, j)
    sgn = nd.elem_sign_lex.astype(np.float64)
    for l, a, b in loc:
        r = nd.elem_dof_lex[:, l].astype(np.int64)
        rows += [r, r]
        cols += [h1.elem_dof_lex[:, a].astype(np.int64), h1.elem_dof_lex[:, b].astype(np.int64)]
        vals += [-sgn[:, l], sgn[:, l]]
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    # one copy per (edge, vertex): all copies are equal
    key = rows * h1.ndofs + cols
    _, first = np.unique(key, return_index=True)
    G = sp.csr_matrix((vals[first], (rows[first], cols[first])), shape=(nd.ndofs, h1.ndofs))
    assert ne == 0 or (np.diff(G.indptr) == 2).all()
    return G


def vertex_coordinates(h1):
    """Coordinates [h1.ndofs, 3] of the dofs of an order-1 H1 space (the mesh vertices in the space's numbering)."""
    from .mesh import _CORNER_LATTICE

    assert h1.p == 1
    xyz = np.zeros((h1.ndofs, 3))
    ec = h1.mesh.elem_coords()  # [ne, 27, 3], lexicographic 3 x 3 x 3 lattice
    # corners of the lattice in the lexicographic order of the order-1 element: (i, j, k) -> lattice (2 i, 2 j, 2 k)
    corner = [2 * i + 3 * (2 * j + 3 * 2 * k) for k in range(2) for j in range(2) for i in range(2)]
    for m, c in enumerate(corner):
        xyz[h1.elem_dof_lex[:, m]] = ec[:, c]
    return xyz
