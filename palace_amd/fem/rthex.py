"""Raviart-Thomas hexahedra and the discrete curl ND -> RT on hex meshes (host-side set-up; see fem/rt.py for what the
reference uses them for: the flux B = curl A, drivers/eigensolver.cpp:469-477, and H(div) mass operators).

Order-p RT hex element, P = 3 p^2 (p + 1), tensor ("lexicographic") ordering with cb = closed Gauss-Lobatto and ob = open
Gauss-Legendre nodal bases (the partners of fem/fespace.py's Nedelec blocks, so that curl ND_p is contained in RT_p):
  x-block: cb_i(x) ob_j(y) ob_k(z) e_x, index i + (p+1) (j + p k)
  y-block: ob_i(x) cb_j(y) ob_k(z) e_y, index i + p (j + (p+1) k)
  z-block: ob_i(x) ob_j(y) cb_k(z) e_z, index i + p (j + p k)
The basis is nodal: dof = the block's Cartesian component at its tensor node, which on a face is the (reference) normal
flux density, invariant under the contravariant Piola map up to the sign of the face normal.
Global dofs = faces | interiors; a face dof carries the sign of the element's reference normal e_n against the normal
s x t of the face's global frame (fem/fespace.py:_face_orientation) - an oriented restriction
(fem/libceed/restriction.cpp:288-298).
"""
from __future__ import annotations

import numpy as np

from .basis1d import gauss_legendre, gauss_lobatto, lagrange_eval
from .fespace import _face_orientation, nd_block_shape
from .mesh import HEX_FACE_AXES, HEX_FACES_UV, HexMesh


def rt_block_shape(p: int, comp: int):
    n = [p, p, p]
    n[comp] = p + 1
    return tuple(n)


def rt_lex_index(p: int, comp: int, i, j, k):
    nx, ny, _ = rt_block_shape(p, comp)
    return comp * p * p * (p + 1) + i + nx * (j + ny * k)


def _nodes(p, closed):
    return gauss_lobatto(p + 1) if closed else gauss_legendre(p)[0]


def rt_hex_tables(p: int, pts1d):
    """interp [3, Q, P], div [Q, P] on the tensor grid of the 1-D points (q = qx + n (qy + n qz))."""
    pts1d = np.asarray(pts1d, dtype=np.float64)
    n = pts1d.size
    Q, P = n ** 3, 3 * p * p * (p + 1)
    Bc, Gc = lagrange_eval(_nodes(p, True), pts1d)
    Bo, _ = lagrange_eval(_nodes(p, False), pts1d)
    interp, div = np.zeros((3, Q, P)), np.zeros((Q, P))
    for comp in range(3):
        B = [Bo, Bo, Bo]
        D = [None, None, None]
        B[comp], D[comp] = Bc, Gc
        nx, ny, nz = rt_block_shape(p, comp)
        val = np.einsum("ck,bj,ai->cbakji", B[2], B[1], B[0]).reshape(Q, nz * ny * nx)
        dB = list(B)
        dB[comp] = D[comp]
        dv = np.einsum("ck,bj,ai->cbakji", dB[2], dB[1], dB[0]).reshape(Q, nz * ny * nx)
        base = comp * p * p * (p + 1)
        interp[comp, :, base:base + nx * ny * nz] = val
        div[:, base:base + nx * ny * nz] = dv
    return interp, div


def hex_curl_matrix(p: int):
    """Element matrix [P_RT, P_ND] (both in tensor order) of the discrete curl: RT dof i of curl(phi_j), in reference
    coordinates (basis.cpp:139-150 asks MFEM's CurlInterpolator for the same matrix)."""
    P_rt, P_nd = 3 * p * p * (p + 1), 3 * p * (p + 1) ** 2
    C = np.zeros((P_rt, P_nd))
    cn, on = _nodes(p, True), _nodes(p, False)
    for rc in range(3):  # RT component: nodes closed along rc, open along the others
        rn = [on, on, on]
        rn[rc] = cn
        rx, ry, rz = rt_block_shape(p, rc)
        rows = (rc * p * p * (p + 1) + np.arange(rx * ry * rz)).reshape(rz, ry, rx)
        for nc in range(3):  # ND component f e_nc: curl = grad f x e_nc, component rc = eps_{rc, d, nc} d_d f
            if nc == rc:
                continue
            d = 3 - rc - nc
            sign = 1.0 if (rc, d, nc) in ((0, 1, 2), (1, 2, 0), (2, 0, 1)) else -1.0
            nn = [cn, cn, cn]
            nn[nc] = on
            tabs = []
            for ax in range(3):
                B, G = lagrange_eval(nn[ax], rn[ax])
                tabs.append(G if ax == d else B)
            nx, ny, nz = nd_block_shape(p, nc)
            blk = sign * np.einsum("ck,bj,ai->cbakji", tabs[2], tabs[1], tabs[0]).reshape(rz * ry * rx, nz * ny * nx)
            C[rows.ravel(), nc * p * (p + 1) ** 2:nc * p * (p + 1) ** 2 + nx * ny * nz] = blk
    return C


class RTHexSpace:
    """Order-p Raviart-Thomas space on a HexMesh: elem_dof_lex [NE, P], elem_sign_lex [NE, P] (+1 / -1)."""

    def __init__(self, mesh: HexMesh, p: int):
        self.mesh, self.p = mesh, p
        self.P = 3 * p * p * (p + 1)
        ne = mesh.ne
        n_f, n_i = p * p, 3 * p * p * (p - 1)
        self.face_base, self.int_base = 0, mesh.nfaces * n_f
        self.ndofs = self.int_base + ne * n_i
        dof = np.full((ne, self.P), -1, dtype=np.int64)
        sgn = np.ones((ne, self.P), dtype=np.int8)
        verts = mesh.verts
        A, B = np.meshgrid(np.arange(p), np.arange(p), indexing="ij")
        A, B = A.ravel(), B.ravel()
        for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
            ou, ov, swap = _face_orientation(verts[:, HEX_FACES_UV[lf]])
            ijk = [None, None, None]
            ijk[nax] = np.full(A.size, side * p)
            ijk[uax], ijk[vax] = A, B
            lex = rt_lex_index(p, nax, ijk[0], ijk[1], ijk[2])
            a2 = np.where(ou[:, None], p - 1 - A[None, :], A[None, :])
            b2 = np.where(ov[:, None], p - 1 - B[None, :], B[None, :])
            g = np.where(swap[:, None], b2 + p * a2, a2 + p * b2)
            dof[:, lex] = self.face_base + mesh.elem_faces[:, lf, None] * n_f + g
            # e_u x e_v = +e_n for (u, v, n) = (x, y, z), (y, z, x); -e_n for (x, z, y); the global frame (s, t) is
            # (+-u, +-v) or, swapped, (+-v, +-u)
            uxv = -1 if nax == 1 else 1
            s = uxv * np.where(ou, -1, 1) * np.where(ov, -1, 1) * np.where(swap, -1, 1)
            sgn[:, lex] = s[:, None]
        cnt = 0
        for comp in range(3):
            rng = [np.arange(p)] * 3
            rng[comp] = np.arange(1, p)
            K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
            lex = rt_lex_index(p, comp, I.ravel(), J.ravel(), K.ravel())
            if lex.size:
                dof[:, lex] = self.int_base + np.arange(ne)[:, None] * n_i + (cnt + np.arange(lex.size))[None, :]
            cnt += lex.size
        assert cnt == n_i and dof.min() >= 0
        self.elem_dof_lex = dof.astype(np.int32)
        self.elem_sign_lex = sgn

    def restriction(self, interp_range=False):
        return dict(offsets=self.elem_dof_lex, lsize=self.ndofs, orients=self.elem_sign_lex < 0)
