"""Raviart-Thomas tetrahedra and the discrete curl (host-side set-up, like fem/tet.py).

What the reference uses these for: the flux B = curl A after every solve
(`Curl.Mult`, drivers/eigensolver.cpp:469-477, built as a DiscreteLinearOperator with the curl interpolator,
fem/bilinearform.cpp:203-282 + basis.cpp:139-150), and H(div) mass operators for the flux error estimator
(linalg/errorestimator.cpp; fem/integ/vecfemass.cpp with f_apply_hdiv_33 for an RT space).

MFEM is not in the reference tree, so the element is restated from its definition:
  * RT space of order p (the partner of the order-p Nedelec space, RT_FECollection(p - 1)):
    P_{p-1}^3 + z P~_{p-1}, dimension p (p + 1) (p + 3) / 2;
  * nodal basis dual to normal-flux point functionals: p (p + 1) / 2 per face (v . n_f at the face's interior
    lattice points, n_f = the reference face's area-weighted outward normal, so the value is invariant under
    the contravariant Piola map), 3 C(p + 1, 3) interior (Cartesian components at interior lattice points);
  * element -> global dofs: faces | interiors; a face dof changes sign when the element's outward normal opposes
    the normal of the face's sorted vertex frame, and its point is matched through the sorted barycentric
    coordinates (an oriented restriction, fem/libceed/restriction.cpp:288-298).
"""
from __future__ import annotations

import numpy as np

from .tet import (LOCAL_FACES, REF_VERTS, TetMesh, _CENTROID, _interior_lattice, _mono_eval, _mono_grad, _monomials,
                  _VecPoly, NDTetElement)


def _div(poly: _VecPoly, x):
    z = x - _CENTROID
    d = np.zeros(z.shape[:-1])
    for c, e, a in poly.terms:
        d += a * _mono_grad(e, z)[..., c]
    return d


def _rt_candidates(p):
    cands = []
    for e in _monomials(0, p - 1):
        for c in range(3):
            cands.append(_VecPoly([(c, e, 1.0)]))
    for e in _monomials(p - 1, p - 1):  # z m(z), m homogeneous of degree p - 1
        terms = []
        for c in range(3):
            ee = list(e)
            ee[c] += 1
            terms.append((c, tuple(ee), 1.0))
        cands.append(_VecPoly(terms))
    return cands


def rt_face_point_bary(p):
    """Barycentric coordinates (wrt the face's three vertices) of the p (p + 1) / 2 face dof points."""
    return _interior_lattice(2, p + 2, p - 1)


def rt_tet_functionals(p):
    """(point, direction) of the dof functionals in local order: faces (LOCAL_FACES order), interior."""
    V = REF_VERTS
    pts, dirs = [], []
    for a, b, c in LOCAL_FACES:
        n = np.cross(V[b] - V[a], V[c] - V[a])  # outward, |n| = 2 area
        for la, lb, lc in rt_face_point_bary(p):
            pts.append(la * V[a] + lb * V[b] + lc * V[c])
            dirs.append(n)
    if p >= 2:
        for l in _interior_lattice(3, p + 2, p - 2):
            x = sum(li * V[i] for i, li in enumerate(l))
            for d in range(3):
                pts.append(x)
                dirs.append(np.eye(3)[d])
    return np.array(pts), np.array(dirs)


class RTTetElement:
    """Order-p Raviart-Thomas tetrahedron: value and divergence tables at arbitrary reference points."""

    def __init__(self, p):
        self.p = p
        self.P = p * (p + 1) * (p + 3) // 2
        self.cands = _rt_candidates(p)
        pts, dirs = rt_tet_functionals(p)
        assert len(pts) == self.P, (len(pts), self.P)
        self.dof_pts, self.dof_dirs = pts, dirs
        # outward normals of the reference faces point away from the opposite vertex
        for k, (a, b, c) in enumerate(LOCAL_FACES):
            opp = ({0, 1, 2, 3} - {a, b, c}).pop()
            n = np.cross(REF_VERTS[b] - REF_VERTS[a], REF_VERTS[c] - REF_VERTS[a])
            assert np.dot(n, REF_VERTS[opp] - REF_VERTS[a]) < 0, "LOCAL_FACES must be outward oriented"
        Vm = np.array([np.einsum("nd,nd->n", c.eval(pts), dirs) for c in self.cands]).T  # [P, ncand]
        self.coef = np.linalg.pinv(Vm, rcond=1e-12)
        assert np.abs(Vm @ self.coef - np.eye(self.P)).max() < 1e-9, "Raviart-Thomas dofs are not unisolvent"

    def tables(self, x):
        """interp [3, Q, P], div [Q, P] at reference points x [Q, 3]."""
        val = np.array([c.eval(x) for c in self.cands])  # [ncand, Q, 3]
        dv = np.array([_div(c, x) for c in self.cands])  # [ncand, Q]
        interp = np.einsum("kqd,kj->dqj", val, self.coef)
        div = np.einsum("kq,kj->qj", dv, self.coef)
        return np.ascontiguousarray(interp), np.ascontiguousarray(div)


def tet_curl_matrix(p):
    """Element matrix [P_RT, P_ND] of the discrete curl ND(p) -> RT(p) in reference coordinates (the curl of an
    H(curl)-mapped field is the H(div)-mapped reference curl, so one matrix serves every element): entry (i, j) =
    RT dof i of curl(phi_j); basis.cpp:139-150 asks MFEM's CurlInterpolator for the same matrix."""
    nd, rt = NDTetElement(p), RTTetElement(p)
    _, curl = nd.tables(rt.dof_pts)  # [3, n_rt, P_nd]
    return np.ascontiguousarray(np.einsum("dij,id->ij", curl, rt.dof_dirs))


class RTTetSpace:
    """Order-p Raviart-Thomas space on a TetMesh: global dofs = faces | interiors, oriented restriction."""

    def __init__(self, mesh: TetMesh, p: int):
        self.mesh, self.p = mesh, p
        self.elem = RTTetElement(p)
        self.P = self.elem.P
        ne = mesh.ne
        n_f = p * (p + 1) // 2
        n_i = self.P - 4 * n_f
        NF = mesh.face_verts.shape[0]
        self.ndofs = NF * n_f + ne * n_i
        self.face_base, self.int_base = 0, NF * n_f
        off = np.zeros((ne, self.P), dtype=np.int64)
        neg = np.zeros((ne, self.P), dtype=bool)
        fb = rt_face_point_bary(p)
        key = {tuple(np.round(np.array(l) * (p + 2)).astype(int)): m for m, l in enumerate(fb)}
        t = mesh.tets
        for k, lf in enumerate(LOCAL_FACES):
            gv = t[:, list(lf)]
            rank = np.argsort(np.argsort(gv, axis=1), axis=1)  # position of local vertex m in the sorted frame
            # parity of the permutation that sorts (A, B, C): odd <=> local normal opposes the sorted frame's
            inv = ((rank[:, 0] > rank[:, 1]).astype(int) + (rank[:, 0] > rank[:, 2]) + (rank[:, 1] > rank[:, 2]))
            flip = (inv % 2) == 1
            gf = mesh.elem_faces[:, k]
            for m, l in enumerate(fb):
                li = np.round(np.array(l) * (p + 2)).astype(int)
                gl = np.zeros((ne, 3), dtype=int)
                for mm in range(3):
                    gl[np.arange(ne), rank[:, mm]] = li[mm]
                gm = np.array([key[tuple(r)] for r in gl])
                off[:, k * n_f + m] = gf * n_f + gm
                neg[:, k * n_f + m] = flip
        for i in range(n_i):
            off[:, 4 * n_f + i] = self.int_base + np.arange(ne) * n_i + i
        self.offsets = off.astype(np.int32)
        self.orients = neg

    def restriction(self, interp_range=False):
        # signs are their own inverse, so the interpolator-range form is the same
        return dict(offsets=self.offsets, lsize=self.ndofs, orients=self.orients)
