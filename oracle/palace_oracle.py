"""CPU oracle (numpy) for Palace's partial-assembly operator apply and its Krylov loop.

*** TEST INFRASTRUCTURE — not product code.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The product path (palace_amd/) never does. ***

What it restates, function by function (all paths relative to the reference tree):
  * libCEED operator semantics `v += E^T B^T D B E u` with *dense* non-tensor H(curl) basis tables,
    exactly what Palace asks libCEED for (palace/fem/libceed/basis.cpp:40-85,
    palace/fem/libceed/restriction.cpp:362-384, palace/fem/libceed/operator.cpp:148-190).
    libCEED itself (pinned 95bd1e908b16e04a70015e3a9a7fddec5e9c3fc8, cmake/ExternalGitTags.cmake:78-79)
    is not vendored; its published semantics for CEED_EVAL_INTERP / CEED_EVAL_CURL on an H(curl)
    basis are `(B u)[d, q] = sum_j interp[(d Q + q) P + j] u_j`.
  * geometry factors  : palace/fem/qfunctions/33/geom_33_qf.h:9-33
  * D, curl-curl      : palace/fem/qfunctions/33/hdiv_33_qf.h:10-30
  * D, mass / H1 diff : palace/fem/qfunctions/33/hcurl_33_qf.h:10-28
  * D, curl-curl+mass : palace/fem/qfunctions/33/hdivmass_33_qf.h:10-44
  * coefficient ctx   : palace/fem/libceed/coefficient.cpp:51-131, palace/fem/qfunctions/coeff/coeff_qf.h
  * ParOperator BCs   : palace/linalg/rap.cpp:195-234, :154-193
  * PCG               : palace/linalg/iterative.cpp:360-486
  * Chebyshev 4th kind: palace/linalg/chebyshev.cpp:160-220;  lambda_max: palace/linalg/operator.cpp:583-631
  * V-cycle           : palace/linalg/gmg.cpp:171-205

Pinning: the D stage is checked against the *real* reference QFunction headers
compiled into oracle/_ref (tests/test_oracle_ref.py, fixtures in tests/golden/); the whole chain
mesh -> basis -> E/B/D -> eigenproblem is pinned on the reference's own regression data
test/data/regression/ref/cylinder/cavity_pec/eig.csv (tests/test_oracle_eigen.py).  The basis
tables and dof numbering that MFEM (not vendored) would supply are NOT pinned entry by entry —
only through those basis-invariant results.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------------------------
# 1-D pieces (independent of palace_amd.fem.basis1d on purpose: different algorithm)
# ---------------------------------------------------------------------------------------------

def gl_points(n):
    """Gauss-Legendre on [0,1] via the Golub-Welsch eigenvalue problem."""
    if n == 1:
        return np.array([0.5]), np.array([1.0])
    k = np.arange(1, n)
    beta = k / np.sqrt(4.0 * k * k - 1.0)
    T = np.diag(beta, 1) + np.diag(beta, -1)
    lam, V = np.linalg.eigh(T)
    w = 2.0 * V[0, :] ** 2
    # polish with Newton on P_n
    x = lam
    for _ in range(3):
        P0, P1 = np.ones_like(x), x
        for m in range(2, n + 1):
            P0, P1 = P1, ((2 * m - 1) * x * P1 - (m - 1) * P0) / m
        dP = n * (x * P1 - P0) / (x * x - 1.0)
        x = x - P1 / dP
    P0, P1 = np.ones_like(x), x
    for m in range(2, n + 1):
        P0, P1 = P1, ((2 * m - 1) * x * P1 - (m - 1) * P0) / m
    dP = n * (x * P1 - P0) / (x * x - 1.0)
    w = 2.0 / ((1.0 - x * x) * dP * dP)
    return 0.5 * (x + 1.0), 0.5 * w


def gll_points(n):
    """Gauss-Lobatto points on [0,1]: eigenvalues of the Jacobi matrix for P'_{n-1} plus ends."""
    if n == 2:
        return np.array([0.0, 1.0])
    m = n - 2  # interior points = roots of P'_{n-1} = Jacobi(1,1) polynomial of degree m
    k = np.arange(1, m)
    beta = np.sqrt(k * (k + 2.0) / ((2.0 * k + 1.0) * (2.0 * k + 3.0)))
    T = np.diag(beta, 1) + np.diag(beta, -1)
    x = np.linalg.eigvalsh(T) if m > 1 else np.array([0.0])
    N = n - 1
    for _ in range(3):  # Newton on P'_N via (1-x^2) P'_N = N (P_{N-1} - x P_N)
        P0, P1 = np.ones_like(x), x
        for j in range(2, N + 1):
            P0, P1 = P1, ((2 * j - 1) * x * P1 - (j - 1) * P0) / j
        f = P0 - x * P1            # proportional to (1-x^2) P'_N
        df = -(N + 1) * P1         # d/dx (P_{N-1} - x P_N) = -(N+1) P_N
        x = x - f / df
    x = np.concatenate([[-1.0], np.sort(x), [1.0]])
    return 0.5 * (x + 1.0)


def lagrange(nodes, x, i):
    """l_i(x), l_i'(x) on `nodes` at scalar x (direct product formulas)."""
    n = len(nodes)
    val, denom = 1.0, 1.0
    for m in range(n):
        if m != i:
            val *= x - nodes[m]
            denom *= nodes[i] - nodes[m]
    der = 0.0
    for m in range(n):
        if m == i:
            continue
        t = 1.0
        for l in range(n):
            if l != i and l != m:
                t *= x - nodes[l]
        der += t
    return val / denom, der / denom


def hex_quadrature(q1d):
    """Tensor Gauss-Legendre rule, point index q = qx + q1d (qy + q1d qz)."""
    x, w = gl_points(q1d)
    pts = np.array([[x[a], x[b], x[c]] for c in range(q1d) for b in range(q1d) for a in range(q1d)])
    wts = np.array([w[a] * w[b] * w[c] for c in range(q1d) for b in range(q1d) for a in range(q1d)])
    return pts, wts


# ---------------------------------------------------------------------------------------------
# Dense reference-element tables, as `fe.GetDofToQuad(ir, DofToQuad::FULL)` gives Palace
# (basis.cpp:43-83): interp[(d*Q+q)*P + j], curl[(d*Q+q)*P + j], native dof order via dof_map.
# ---------------------------------------------------------------------------------------------

def nd_hex_dense_tables(p, q1d, dof_map):
    """Direct point-wise evaluation of every Nedelec hex shape function and its curl.

    dof_map[lex] = native index (or -1-native for a flipped shape function)."""
    cp, op = gll_points(p + 1), gl_points(p)[0]
    pts, _ = hex_quadrature(q1d)
    Q, P = pts.shape[0], 3 * p * (p + 1) ** 2
    interp = np.zeros((3, Q, P))
    curl = np.zeros((3, Q, P))
    for comp in range(3):
        n = [p + 1] * 3
        n[comp] = p
        nodes = [cp, cp, cp]
        nodes[comp] = op
        for k in range(n[2]):
            for j in range(n[1]):
                for i in range(n[0]):
                    lex = comp * p * (p + 1) ** 2 + i + n[0] * (j + n[1] * k)
                    nat = dof_map[lex]
                    s = 1.0
                    if nat < 0:
                        nat, s = -1 - nat, -1.0
                    for q in range(Q):
                        vx, dx = lagrange(nodes[0], pts[q, 0], i)
                        vy, dy = lagrange(nodes[1], pts[q, 1], j)
                        vz, dz = lagrange(nodes[2], pts[q, 2], k)
                        val = vx * vy * vz
                        grad = (dx * vy * vz, vx * dy * vz, vx * vy * dz)
                        interp[comp, q, nat] = s * val
                        # curl(f e_c): components (e_a x ...)
                        if comp == 0:   # f e_x -> (0, df/dz, -df/dy)
                            curl[1, q, nat], curl[2, q, nat] = s * grad[2], -s * grad[1]
                        elif comp == 1:  # f e_y -> (-df/dz, 0, df/dx)
                            curl[0, q, nat], curl[2, q, nat] = -s * grad[2], s * grad[0]
                        else:           # f e_z -> (df/dy, -df/dx, 0)
                            curl[0, q, nat], curl[1, q, nat] = s * grad[1], -s * grad[0]
    return interp.reshape(-1), curl.reshape(-1)


def h1_hex_dense_tables(p, q1d):
    """H1 tensor element, lexicographic dofs: interp[q*P + j], grad[(d*Q+q)*P + j]."""
    cp = gll_points(p + 1)
    pts, _ = hex_quadrature(q1d)
    n1 = p + 1
    Q, P = pts.shape[0], n1**3
    interp = np.zeros((Q, P))
    grad = np.zeros((3, Q, P))
    for k in range(n1):
        for j in range(n1):
            for i in range(n1):
                lex = i + n1 * (j + n1 * k)
                for q in range(Q):
                    vx, dx = lagrange(cp, pts[q, 0], i)
                    vy, dy = lagrange(cp, pts[q, 1], j)
                    vz, dz = lagrange(cp, pts[q, 2], k)
                    interp[q, lex] = vx * vy * vz
                    grad[0, q, lex] = dx * vy * vz
                    grad[1, q, lex] = vx * dy * vz
                    grad[2, q, lex] = vx * vy * dz
    return interp.reshape(-1), grad.reshape(-1)


def mesh_q2_grad_table(q1d):
    """grad table of the tri-quadratic mesh-node basis at the quadrature points:
    G[d, q, n] for lattice node n = i + 3 j + 9 k (nodes at {0, 1/2, 1}^3)."""
    nodes = np.array([0.0, 0.5, 1.0])
    pts, _ = hex_quadrature(q1d)
    Q = pts.shape[0]
    G = np.zeros((3, Q, 27))
    for k in range(3):
        for j in range(3):
            for i in range(3):
                n = i + 3 * j + 9 * k
                for q in range(Q):
                    vx, dx = lagrange(nodes, pts[q, 0], i)
                    vy, dy = lagrange(nodes, pts[q, 1], j)
                    vz, dz = lagrange(nodes, pts[q, 2], k)
                    G[0, q, n], G[1, q, n], G[2, q, n] = dx * vy * vz, vx * dy * vz, vx * vy * dz
    return G


# ---------------------------------------------------------------------------------------------
# QFunctions (column-major 3x3: M[i + 3 j] = M_ij)
# ---------------------------------------------------------------------------------------------

def adjJt33(J):
    """utils_33_qf.h:20-37.  J: [..., 9] column-major.  Returns (adj(J)^T, det J)."""
    A = np.empty_like(J)
    A[..., 0] = J[..., 4] * J[..., 8] - J[..., 7] * J[..., 5]
    A[..., 3] = J[..., 7] * J[..., 2] - J[..., 1] * J[..., 8]
    A[..., 6] = J[..., 1] * J[..., 5] - J[..., 4] * J[..., 2]
    A[..., 1] = J[..., 6] * J[..., 5] - J[..., 3] * J[..., 8]
    A[..., 4] = J[..., 0] * J[..., 8] - J[..., 6] * J[..., 2]
    A[..., 7] = J[..., 3] * J[..., 2] - J[..., 0] * J[..., 5]
    A[..., 2] = J[..., 3] * J[..., 7] - J[..., 6] * J[..., 4]
    A[..., 5] = J[..., 6] * J[..., 1] - J[..., 0] * J[..., 7]
    A[..., 8] = J[..., 0] * J[..., 4] - J[..., 3] * J[..., 1]
    det = J[..., 0] * A[..., 0] + J[..., 1] * A[..., 1] + J[..., 2] * A[..., 2]
    return A, det


def mult_AtBCx33(A, B, C, x):
    """utils_33_qf.h:64-84: y = A^T B C x, matrices [..., 9] column-major, x [..., 3]."""
    y0 = C[..., 0] * x[..., 0] + C[..., 3] * x[..., 1] + C[..., 6] * x[..., 2]
    y1 = C[..., 1] * x[..., 0] + C[..., 4] * x[..., 1] + C[..., 7] * x[..., 2]
    y2 = C[..., 2] * x[..., 0] + C[..., 5] * x[..., 1] + C[..., 8] * x[..., 2]
    z0 = B[..., 0] * y0 + B[..., 3] * y1 + B[..., 6] * y2
    z1 = B[..., 1] * y0 + B[..., 4] * y1 + B[..., 7] * y2
    z2 = B[..., 2] * y0 + B[..., 5] * y1 + B[..., 8] * y2
    out = np.empty(np.broadcast_shapes(z0.shape, A[..., 0].shape) + (3,))
    out[..., 0] = A[..., 0] * z0 + A[..., 1] * z1 + A[..., 2] * z2
    out[..., 1] = A[..., 3] * z0 + A[..., 4] * z1 + A[..., 5] * z2
    out[..., 2] = A[..., 6] * z0 + A[..., 7] * z1 + A[..., 8] * z2
    return out


def build_geom_factor_33(attr, qw, J):
    """geom_33_qf.h:9-33.  attr [NE], qw [Q], J [NE, Q, 9] column-major (J[i+3j] = dx_i/dxi_j).
    Returns geom [NE, 11, Q]: attr, w detJ, adj(J)^T / detJ."""
    NE, Q = J.shape[0], J.shape[1]
    A, det = adjJt33(J)
    geom = np.empty((NE, 11, Q))
    geom[:, 0, :] = attr[:, None]
    geom[:, 1, :] = qw[None, :] * det
    geom[:, 2:, :] = np.transpose(A / det[..., None], (0, 2, 1))
    return geom


class CoeffCtx:
    """CeedIntScalar context (coeff_qf.h:7-45; packing coefficient.cpp:51-118):
    [nattr][attr->mat (nattr)][nmat][nmat*dim*dim doubles col-major]; 8-byte slots.
    dim = 3 for matrix coefficients, 1 for the scalar ones of the H1 mass QFunctions."""

    def __init__(self, attr_mat=None, mat_coeff=None, a=1.0, dim=3):
        self.dim = dim
        d2 = dim * dim
        if attr_mat is None:  # no coefficient: identity scaled by a (coefficient.cpp:55-64)
            self.attr_mat = np.zeros(0, dtype=np.int32)
            self.mat = (a * np.eye(dim)).reshape(1, d2)
        else:
            attr_mat = np.asarray(attr_mat, dtype=np.int32)
            mats = [np.asarray(m, dtype=np.float64) for m in mat_coeff]
            nmat = len(mats)
            full = np.zeros((nmat + 1, d2))
            for k, mk in enumerate(mats):
                if mk.size == 1:
                    full[k] = (a * float(mk.reshape(-1)[0]) * np.eye(dim)).reshape(-1)
                else:
                    full[k] = (a * mk).reshape(dim, dim).T.reshape(-1)  # column-major
            self.attr_mat = np.where(attr_mat < 0, nmat, attr_mat).astype(np.int32)
            self.mat = full

    def pack(self) -> np.ndarray:
        """The raw blob as 8-byte slots (ints in the low 4 bytes), returned as float64 view."""
        nattr, nmat = self.attr_mat.size, self.mat.shape[0]
        raw = np.zeros(2 + nattr + self.mat.size, dtype=np.float64)
        iv = raw.view(np.int32).reshape(-1, 2)
        iv[0, 0] = nattr
        iv[1 : 1 + nattr, 0] = self.attr_mat
        iv[1 + nattr, 0] = nmat
        raw[2 + nattr :] = self.mat.reshape(-1)
        return raw

    def unpack3(self, attr):
        """CoeffUnpack3 / CoeffUnpack1 (coeff_3_qf.h:9-24, coeff_1_qf.h): attr (1-based) -> [..., dim*dim]."""
        if self.attr_mat.size > 0:
            k = self.attr_mat[attr - 1]
        else:
            k = np.zeros_like(attr)
        return self.mat[k]


def pack_pair(ctx_mass: CoeffCtx, ctx: CoeffCtx) -> np.ndarray:
    """coefficient.cpp:120-131: mass context first."""
    return np.concatenate([ctx_mass.pack(), ctx.pack()])


def apply_hcurl_33(ctx: CoeffCtx, geom, u):
    """hcurl_33_qf.h:10-28.  geom [NE, 11, Q], u [NE, 3, Q] -> v [NE, 3, Q]."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))  # [NE, Q, 9]
    C = ctx.unpack3(attr)
    v = mult_AtBCx33(adj, C, adj, np.transpose(u, (0, 2, 1)))
    return np.transpose(wdetJ[..., None] * v, (0, 2, 1))


def apply_hdiv_33(ctx: CoeffCtx, geom, u):
    """hdiv_33_qf.h:10-30: J/detJ = adj(adjJt)^T recomputed per point."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))
    Jl, _ = adjJt33(adj)
    C = ctx.unpack3(attr)
    v = mult_AtBCx33(Jl, C, Jl, np.transpose(u, (0, 2, 1)))
    return np.transpose(wdetJ[..., None] * v, (0, 2, 1))


def apply_hcurlhdiv_33(ctx: CoeffCtx, geom, u):
    """hcurlhdiv_33_qf.h:10-31 (f_apply_hcurlhdiv_33): v = w detJ (J/detJ)^T C adjJt u -- H(curl) values in, H(div)
    (curl) test functions out: the weak curl (C u, curl v) of MixedVectorWeakCurlIntegrator (integ/mixedveccurl.cpp:75-120)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))
    Jl, _ = adjJt33(adj)
    v = mult_AtBCx33(Jl, ctx.unpack3(attr), adj, np.transpose(u, (0, 2, 1)))
    return np.transpose(wdetJ[..., None] * v, (0, 2, 1))


def apply_hdivhcurl_33(ctx: CoeffCtx, geom, u):
    """hcurlhdiv_33_qf.h:33-54 (f_apply_hdivhcurl_33): v = w detJ adjJt^T C (J/detJ) u -- curls in, H(curl) test values
    out: (C curl u, v) of MixedVectorCurlIntegrator (integ/mixedveccurl.cpp:21-73)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))
    Jl, _ = adjJt33(adj)
    v = mult_AtBCx33(adj, ctx.unpack3(attr), Jl, np.transpose(u, (0, 2, 1)))
    return np.transpose(wdetJ[..., None] * v, (0, 2, 1))


def mult_BAx33(A, B, x):
    """utils_33_qf.h:86-101: y = B (A x), matrices [..., 9] column-major, x [..., 3]."""
    z0 = A[..., 0] * x[..., 0] + A[..., 3] * x[..., 1] + A[..., 6] * x[..., 2]
    z1 = A[..., 1] * x[..., 0] + A[..., 4] * x[..., 1] + A[..., 7] * x[..., 2]
    z2 = A[..., 2] * x[..., 0] + A[..., 5] * x[..., 1] + A[..., 8] * x[..., 2]
    out = np.empty(np.broadcast_shapes(z0.shape, B[..., 0].shape) + (3,))
    out[..., 0] = B[..., 0] * z0 + B[..., 3] * z1 + B[..., 6] * z2
    out[..., 1] = B[..., 1] * z0 + B[..., 4] * z1 + B[..., 7] * z2
    out[..., 2] = B[..., 2] * z0 + B[..., 5] * z1 + B[..., 8] * z2
    return out


def apply_hcurlhdiv_error_33(ctx1: CoeffCtx, ctx2: CoeffCtx, geom, u1, u2):
    """hcurlhdiv_error_33_qf.h:10-43 (f_apply_hcurlhdiv_error_33): u1 [NE, 3, Q] reference values of an H(curl) function,
    u2 of an H(div) function; returns w detJ |C2 (J/detJ) u2 - C1 adjJt u1|^2  [NE, Q] (the integrand of the element error of
    GradFluxErrorEstimator, linalg/errorestimator.cpp:318-349)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))
    Jl, _ = adjJt33(adj)
    v1 = mult_BAx33(adj, ctx1.unpack3(attr), np.transpose(u1, (0, 2, 1)))
    v2 = mult_BAx33(Jl, ctx2.unpack3(attr), np.transpose(u2, (0, 2, 1)))
    d = v2 - v1
    return wdetJ * (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])


def apply_hdivhcurl_error_33(ctx1: CoeffCtx, ctx2: CoeffCtx, geom, u1, u2):
    """hcurlhdiv_error_33_qf.h:45-78 (f_apply_hdivhcurl_error_33): the first input in H(div), the second in H(curl)
    (CurlFluxErrorEstimator, linalg/errorestimator.cpp:448-489)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    adj = np.transpose(geom[:, 2:, :], (0, 2, 1))
    Jl, _ = adjJt33(adj)
    v1 = mult_BAx33(Jl, ctx1.unpack3(attr), np.transpose(u1, (0, 2, 1)))
    v2 = mult_BAx33(adj, ctx2.unpack3(attr), np.transpose(u2, (0, 2, 1)))
    d = v2 - v1
    return wdetJ * (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])


def apply_hdivmass_33(ctx_mass: CoeffCtx, ctx_curl: CoeffCtx, geom, u, curlu):
    """hdivmass_33_qf.h:10-44 (mass coefficient first, then the curl-curl one)."""
    return apply_hcurl_33(ctx_mass, geom, u), apply_hdiv_33(ctx_curl, geom, curlu)


# ---- 2-D (space_dim = dim = 2) variants: fem/qfunctions/22/*.h, fem/qfunctions/1/l2_1_qf.h --------

def build_geom_factor_22(attr, qw, J):
    """geom_22_qf.h:9-30.  J [NE, Q, 4] column-major (J: 0 2 / 1 3).  Returns geom [NE, 6, Q]:
    attr, w detJ, adj(J)^T / detJ (utils_22_qf.h:20-31: {J3, -J2, -J1, J0} / det)."""
    det = J[..., 0] * J[..., 3] - J[..., 1] * J[..., 2]
    NE, Q = det.shape
    geom = np.empty((NE, 6, Q))
    geom[:, 0, :] = attr[:, None]
    geom[:, 1, :] = qw[None, :] * det
    geom[:, 2, :] = J[..., 3] / det
    geom[:, 3, :] = -J[..., 2] / det
    geom[:, 4, :] = -J[..., 1] / det
    geom[:, 5, :] = J[..., 0] / det
    return geom


def _unpack2(ctx, attr):
    k = ctx.attr_mat[attr - 1] if ctx.attr_mat.size else np.zeros_like(attr)
    return ctx.mat[k]  # [..., 4] column-major


def _unpack1(ctx, attr):
    k = ctx.attr_mat[attr - 1] if ctx.attr_mat.size else np.zeros_like(attr)
    return ctx.mat[k][..., 0]


def apply_hcurl_22(ctx, geom, u):
    """hcurl_22_qf.h:10-30: v = w detJ A^T C A u with A = adjJt (utils_22_qf.h MultAtBCx22)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    A = [geom[:, 2 + k, :] for k in range(4)]
    C = _unpack2(ctx, attr)
    x0, x1 = u[:, 0, :], u[:, 1, :]
    y0 = A[0] * x0 + A[2] * x1
    y1 = A[1] * x0 + A[3] * x1
    z0 = C[..., 0] * y0 + C[..., 2] * y1
    z1 = C[..., 1] * y0 + C[..., 3] * y1
    return np.stack([wdetJ * (A[0] * z0 + A[1] * z1), wdetJ * (A[2] * z0 + A[3] * z1)], axis=1)


def _mult_AtBCx22(A, B, C, x0, x1):
    """utils_22_qf.h:49-65: y = A^T B C x, matrices as lists / [..., 4] column-major."""
    y0 = C[0] * x0 + C[2] * x1
    y1 = C[1] * x0 + C[3] * x1
    z0 = B[0] * y0 + B[2] * y1
    z1 = B[1] * y0 + B[3] * y1
    return A[0] * z0 + A[1] * z1, A[2] * z0 + A[3] * z1


def _mult_BAx22(A, B, x0, x1):
    """utils_22_qf.h:67-79: y = B (A x)."""
    z0 = A[0] * x0 + A[2] * x1
    z1 = A[1] * x0 + A[3] * x1
    return B[0] * z0 + B[2] * z1, B[1] * z0 + B[3] * z1


def _geom22(ctx, geom):
    attr = geom[:, 0, :].astype(np.int32)
    A = [geom[:, 2 + k, :] for k in range(4)]
    Jl = [A[3], -A[2], -A[1], A[0]]  # AdjJt22 of adjJt (utils_22_qf.h:19-29): J / detJ
    Cm = _unpack2(ctx, attr)
    return geom[:, 1, :], A, Jl, [Cm[..., k] for k in range(4)]


def apply_l2h1_error(ctx1, ctx2, geom, u1, u2):
    """l2h1_error_qf.h:14-30 (f_apply_l2h1_error): w detJ (c1 u1 - c2 u2)^2 for two scalar fields [NE, 1, Q] -> [NE, Q]; any
    geometry data (only rows 0, 1 are read)."""
    attr = geom[:, 0, :].astype(np.int32)
    d = _unpack1(ctx1, attr) * u1[:, 0, :] - _unpack1(ctx2, attr) * u2[:, 0, :]
    return geom[:, 1, :] * d * d


def apply_h1_1(ctx, geom, u):
    """h1_1_qf.h:10-24 (f_apply_h1_1): v = c w detJ u, scalar."""
    attr = geom[:, 0, :].astype(np.int32)
    return (_unpack1(ctx, attr) * geom[:, 1, :])[:, None, :] * u


def apply_hdiv_22(ctx, geom, u):
    """hdiv_22_qf.h:10-30 (f_apply_hdiv_22): v = w detJ (J/detJ)^T C (J/detJ) u -- mass of a plane H(div) space."""
    wdetJ, A, Jl, C = _geom22(ctx, geom)
    v0, v1 = _mult_AtBCx22(Jl, C, Jl, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def apply_hcurlhdiv_22(ctx, geom, u):
    """hcurlhdiv_22_qf.h:10-30 (f_apply_hcurlhdiv_22): v = w detJ (J/detJ)^T C adjJt u."""
    wdetJ, A, Jl, C = _geom22(ctx, geom)
    v0, v1 = _mult_AtBCx22(Jl, C, A, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def apply_hdivhcurl_22(ctx, geom, u):
    """hcurlhdiv_22_qf.h:32-52 (f_apply_hdivhcurl_22): v = w detJ adjJt^T C (J/detJ) u."""
    wdetJ, A, Jl, C = _geom22(ctx, geom)
    v0, v1 = _mult_AtBCx22(A, C, Jl, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def apply_hcurlhdiv_error_22(ctx1, ctx2, geom, u1, u2):
    """hcurlhdiv_error_22_qf.h:10-41: w detJ |C2 (J/detJ) u2 - C1 adjJt u1|^2 with u1 in H(curl), u2 in H(div)  [NE, Q]."""
    wdetJ, A, Jl, C1 = _geom22(ctx1, geom)
    C2 = _geom22(ctx2, geom)[3]
    a0, a1 = _mult_BAx22(A, C1, u1[:, 0, :], u1[:, 1, :])
    b0, b1 = _mult_BAx22(Jl, C2, u2[:, 0, :], u2[:, 1, :])
    b0, b1 = b0 - a0, b1 - a1
    return wdetJ * (b0 * b0 + b1 * b1)


def apply_hdivhcurl_error_22(ctx1, ctx2, geom, u1, u2):
    """hcurlhdiv_error_22_qf.h:43-74: the first input in H(div), the second in H(curl)."""
    wdetJ, A, Jl, C1 = _geom22(ctx1, geom)
    C2 = _geom22(ctx2, geom)[3]
    a0, a1 = _mult_BAx22(Jl, C1, u1[:, 0, :], u1[:, 1, :])
    b0, b1 = _mult_BAx22(A, C2, u2[:, 0, :], u2[:, 1, :])
    b0, b1 = b0 - a0, b1 - a1
    return wdetJ * (b0 * b0 + b1 * b1)


def apply_l2_1(ctx, geom, qw, u):
    """l2_1_qf.h:10-24 (2-D curl-curl, integ/curlcurl.cpp:43-47,65-68): v = (c qw^2 / w detJ) u, scalar."""
    attr = geom[:, 0, :].astype(np.int32)
    return (_unpack1(ctx, attr) * qw[None, :] ** 2 / geom[:, 1, :])[:, None, :] * u


def apply_hdivmass_22(ctx_mass, ctx_curl, geom, qw, u, curlu):
    """hdivmass_22_qf.h:11-37 (mass context, dim 2, first; then the scalar curl-curl one)."""
    return apply_hcurl_22(ctx_mass, geom, u), apply_l2_1(ctx_curl, geom, qw, curlu)


# ---- boundary elements (dim = 2 in space_dim = 3): fem/qfunctions/32/*.h --------------------------------

def build_geom_factor_32(attr, qw, J):
    """geom_32_qf.h:9-33, utils_32_qf.h:23-40.  J [NE, Q, 6] column-major 3x2 (J: 0 3 / 1 4 / 2 5).
    Returns geom [NE, 8, Q]: attr, w detJ (detJ = sqrt(EG - F^2)), adj(J)^T / detJ (3x2)."""
    E = J[..., 0] ** 2 + J[..., 1] ** 2 + J[..., 2] ** 2
    G = J[..., 3] ** 2 + J[..., 4] ** 2 + J[..., 5] ** 2
    F = J[..., 0] * J[..., 3] + J[..., 1] * J[..., 4] + J[..., 2] * J[..., 5]
    d = np.sqrt(E * G - F * F)
    NE, Q = d.shape
    geom = np.empty((NE, 8, Q))
    geom[:, 0, :] = attr[:, None]
    geom[:, 1, :] = qw[None, :] * d
    for k in range(3):
        geom[:, 2 + k, :] = (G * J[..., k] - F * J[..., 3 + k]) / d / d
        geom[:, 5 + k, :] = (E * J[..., 3 + k] - F * J[..., k]) / d / d
    return geom


def apply_hcurl_32(ctx, geom, u):
    """hcurl_32_qf.h:10-30: v = w detJ A^T C A u, A = adjJt (3x2), C 3x3 (utils_32_qf.h:53-72)."""
    attr = geom[:, 0, :].astype(np.int32)
    wdetJ = geom[:, 1, :]
    A = [geom[:, 2 + k, :] for k in range(6)]
    C = ctx.unpack3(attr)
    x0, x1 = u[:, 0, :], u[:, 1, :]
    y = [A[0] * x0 + A[3] * x1, A[1] * x0 + A[4] * x1, A[2] * x0 + A[5] * x1]
    z = [C[..., 0 + r] * y[0] + C[..., 3 + r] * y[1] + C[..., 6 + r] * y[2] for r in range(3)]
    return np.stack([wdetJ * (A[0] * z[0] + A[1] * z[1] + A[2] * z[2]),
                     wdetJ * (A[3] * z[0] + A[4] * z[1] + A[5] * z[2])], axis=1)


# ---- line elements (dim = 1 in space_dim = 2 or 3): fem/qfunctions/21/*.h, fem/qfunctions/31/*.h ----------------------

def build_geom_factor_line(attr, qw, J):
    """geom_21_qf.h:9-30 / geom_31_qf.h:9-31.  J [NE, Q, sdim] (the tangent dx/dxi).  Returns geom [NE, 2 + sdim, Q]:
    attr, w |J|, adj(J)^T / detJ = J / |J|^2 (utils_21_qf.h:20-31)."""
    d = np.sqrt((J * J).sum(axis=-1))
    NE, Q, sdim = J.shape
    geom = np.empty((NE, 2 + sdim, Q))
    geom[:, 0, :] = attr[:, None]
    geom[:, 1, :] = qw[None, :] * d
    for i in range(sdim):
        geom[:, 2 + i, :] = (J[..., i] / d) / d
    return geom


def apply_hcurl_line(ctx, geom, u):
    """hcurl_21_qf.h:10-29 / hcurl_31_qf.h: v = w detJ a^T C a u with a = adjJt (sdim entries), C sdim x sdim column-major
    (MultAtBCx21 / MultAtBCx31, utils_31_qf.h:41-59).  u, v [NE, 1, Q]."""
    sdim = geom.shape[1] - 2
    attr = geom[:, 0, :].astype(np.int32)
    a = [geom[:, 2 + i, :] for i in range(sdim)]
    C = (_unpack2(ctx, attr) if sdim == 2 else ctx.unpack3(attr))
    s = 0.0
    for i in range(sdim):
        z = sum(C[..., i + sdim * j] * a[j] for j in range(sdim))
        s = s + a[i] * z
    return (geom[:, 1, :] * s)[:, None, :] * u


def apply_hcurlmass_line(ctx_mass, ctx, geom, u, gradu):
    """hcurlmass_21_qf.h / hcurlmass_31_qf.h:10-38: H1 mass c w detJ u (first context, dim 1) + the line form above on du/dxi."""
    attr = geom[:, 0, :].astype(np.int32)
    return (_unpack1(ctx_mass, attr) * geom[:, 1, :])[:, None, :] * u, apply_hcurl_line(ctx, geom, gradu)


def apply_hdivmass_32(ctx_mass, ctx_curl, geom, qw, u, curlu):
    """hdivmass_32_qf.h:11-42: ND mass on a boundary element (3x3 material, first context) + the scalar surface curl-curl
    c qw^2 / (w detJ) (second context, dim 1)."""
    return apply_hcurl_32(ctx_mass, geom, u), apply_l2_1(ctx_curl, geom, qw, curlu)


def apply_hcurlmass_22(ctx_mass, ctx, geom, u, gradu):
    """hcurlmass_22_qf.h:13-38: H1 mass c w detJ u (first context, dim 1) + diffusion on grad u (second, 2x2)."""
    attr = geom[:, 0, :].astype(np.int32)
    return (_unpack1(ctx_mass, attr) * geom[:, 1, :])[:, None, :] * u, apply_hcurl_22(ctx, geom, gradu)


def apply_hcurlmass_32(ctx_mass, ctx, geom, u, gradu):
    """hcurlmass_32_qf.h:11-38: the same on a boundary element (3x3 material for the diffusion part)."""
    attr = geom[:, 0, :].astype(np.int32)
    return (_unpack1(ctx_mass, attr) * geom[:, 1, :])[:, None, :] * u, apply_hcurl_32(ctx, geom, gradu)


# ---- the contravariant (H(div)) members on boundary and line elements, the div-div + mass pairs, the gradient form ------

def _adjJt32(A):
    """utils_32_qf.h:23-40 (AdjJt32) applied to the stored adj(J)^T / detJ (six arrays, 3 x 2 column-major): the matrix the
    `hdiv_32` family maps H(div) values with."""
    E = A[0] * A[0] + A[1] * A[1] + A[2] * A[2]
    G = A[3] * A[3] + A[4] * A[4] + A[5] * A[5]
    F = A[0] * A[3] + A[1] * A[4] + A[2] * A[5]
    d = np.sqrt(E * G - F * F)
    return [(G * A[k] - F * A[3 + k]) / d for k in range(3)] + [(E * A[3 + k] - F * A[k]) / d for k in range(3)]


def _mult_AtBCx32(A, B, C, x0, x1):
    """utils_32_qf.h:53-72: y = A^T B C x; A, C 3 x 2 (six arrays), B [..., 9] column-major."""
    y = [C[0] * x0 + C[3] * x1, C[1] * x0 + C[4] * x1, C[2] * x0 + C[5] * x1]
    z = [B[..., 0 + r] * y[0] + B[..., 3 + r] * y[1] + B[..., 6 + r] * y[2] for r in range(3)]
    return A[0] * z[0] + A[1] * z[1] + A[2] * z[2], A[3] * z[0] + A[4] * z[1] + A[5] * z[2]


def _geom32(ctx, geom):
    attr = geom[:, 0, :].astype(np.int32)
    A = [geom[:, 2 + k, :] for k in range(6)]
    return geom[:, 1, :], A, _adjJt32(A), ctx.unpack3(attr)


def apply_hdiv_32(ctx, geom, u):
    """hdiv_32_qf.h:10-31 (f_apply_hdiv_32): v = w detJ Jl^T C Jl u, Jl = AdjJt32(adjJt) -- H(div) mass on boundary elements."""
    wdetJ, A, Jl, C = _geom32(ctx, geom)
    v0, v1 = _mult_AtBCx32(Jl, C, Jl, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def apply_hcurlhdiv_32(ctx, geom, u):
    """hcurlhdiv_32_qf.h:10-30 (f_apply_hcurlhdiv_32): v = w detJ Jl^T C adjJt u."""
    wdetJ, A, Jl, C = _geom32(ctx, geom)
    v0, v1 = _mult_AtBCx32(Jl, C, A, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def apply_hdivhcurl_32(ctx, geom, u):
    """hcurlhdiv_32_qf.h:32-52 (f_apply_hdivhcurl_32): v = w detJ adjJt^T C Jl u."""
    wdetJ, A, Jl, C = _geom32(ctx, geom)
    v0, v1 = _mult_AtBCx32(A, C, Jl, u[:, 0, :], u[:, 1, :])
    return np.stack([wdetJ * v0, wdetJ * v1], axis=1)


def _geom_line(ctx, geom):
    """Line elements: (w detJ, a = adjJt, AdjJt21 / AdjJt31 of it = a / |a| (utils_21_qf.h:19-28, utils_31_qf.h:19-31), C)."""
    sdim = geom.shape[1] - 2
    attr = geom[:, 0, :].astype(np.int32)
    a = [geom[:, 2 + i, :] for i in range(sdim)]
    d = np.sqrt(sum(ai * ai for ai in a))
    C = (_unpack2(ctx, attr) if sdim == 2 else ctx.unpack3(attr))
    return sdim, geom[:, 1, :], a, [ai / d for ai in a], C


def _line_form(sdim, L, C, R):
    """MultAtBCx21 / MultAtBCx31 with x = 1 (utils_21_qf.h, utils_31_qf.h:41-59): L^T C R for column vectors L, R."""
    s = 0.0
    for i in range(sdim):
        s = s + L[i] * sum(C[..., i + sdim * j] * R[j] for j in range(sdim))
    return s


def apply_hdiv_line(ctx, geom, u):
    """hdiv_21_qf.h / hdiv_31_qf.h:10-28 (f_apply_hdiv_21 | _31): v = w detJ t^T C t u with t = AdjJt(adjJt)."""
    sdim, wdetJ, a, t, C = _geom_line(ctx, geom)
    return (wdetJ * _line_form(sdim, t, C, t))[:, None, :] * u


def apply_hcurlhdiv_line(ctx, geom, u):
    """hcurlhdiv_21_qf.h / hcurlhdiv_31_qf.h:10-28 (f_apply_hcurlhdiv_21 | _31): v = w detJ t^T C adjJt u."""
    sdim, wdetJ, a, t, C = _geom_line(ctx, geom)
    return (wdetJ * _line_form(sdim, t, C, a))[:, None, :] * u


def apply_hdivhcurl_line(ctx, geom, u):
    """hcurlhdiv_21_qf.h / hcurlhdiv_31_qf.h:30-48 (f_apply_hdivhcurl_21 | _31): v = w detJ adjJt^T C t u."""
    sdim, wdetJ, a, t, C = _geom_line(ctx, geom)
    return (wdetJ * _line_form(sdim, a, C, t))[:, None, :] * u


def apply_l2mass(ctx_mass, ctx_div, geom, qw, u, divu):
    """l2mass_{22,33,21,31,32}_qf.h (f_apply_l2mass_*): the H(div) mass of the geometry (first context, space_dim x space_dim) on
    the values + c qw^2 / (w detJ) on the divergence (second context, scalar): DivDivMassIntegrator, integ/divdivmass.cpp."""
    rows = geom.shape[1]
    mass = {11: apply_hdiv_33, 6: apply_hdiv_22, 8: apply_hdiv_32, 4: apply_hdiv_line, 5: apply_hdiv_line}[rows]
    return mass(ctx_mass, geom, u), apply_l2_1(ctx_div, geom, qw, divu)


def apply_h1_vec(ctx, geom, u):
    """h1_2_qf.h / h1_3_qf.h:10-33 (f_apply_h1_2 | _3): v = w detJ C u for the N = 2 | 3 components of a vector-valued scalar
    space, C an N x N coefficient (MassIntegrator with num_comp components, integ/mass.cpp:35-48); any geometry data."""
    n = u.shape[1]
    attr = geom[:, 0, :].astype(np.int32)
    C = _unpack2(ctx, attr) if n == 2 else ctx.unpack3(attr)
    return np.stack([geom[:, 1, :] * sum(C[..., i + n * j] * u[:, j, :] for j in range(n)) for i in range(n)], axis=1)


def apply_l2_vec(ctx, geom, qw, u):
    """l2_2_qf.h / l2_3_qf.h:10-34 (f_apply_l2_2 | _3): v = (qw^2 / w detJ) C u, the same with the weight of the scalar-derivative
    forms (DivDivIntegrator with num_comp components, integ/divdiv.cpp:36-47; no element of the hot path has such a divergence)."""
    n = u.shape[1]
    attr = geom[:, 0, :].astype(np.int32)
    C = _unpack2(ctx, attr) if n == 2 else ctx.unpack3(attr)
    w = qw[None, :] ** 2 / geom[:, 1, :]
    return np.stack([w * sum(C[..., i + n * j] * u[:, j, :] for j in range(n)) for i in range(n)], axis=1)


def apply_hcurlh1d(ctx, geom, u):
    """hcurlh1d_{22,33,21,31,32}_qf.h (f_apply_hcurlh1d_*): v = w detJ C (adjJt u) -- reference gradient (dim components) in,
    space_dim physical components out (MultBAx*): GradientIntegrator, integ/grad.cpp:16-72 (trial Grad, test Interp on a
    vector H1 space)."""
    rows = geom.shape[1]
    sdim, dim = {11: (3, 3), 6: (2, 2), 8: (3, 2), 4: (2, 1), 5: (3, 1)}[rows]
    attr = geom[:, 0, :].astype(np.int32)
    A = [geom[:, 2 + k, :] for k in range(sdim * dim)]  # column-major sdim x dim
    C = _unpack2(ctx, attr) if sdim == 2 else ctx.unpack3(attr)
    z = [sum(A[i + sdim * j] * u[:, j, :] for j in range(dim)) for i in range(sdim)]
    return np.stack([geom[:, 1, :] * sum(C[..., i + sdim * j] * z[j] for j in range(sdim)) for i in range(sdim)], axis=1)


# ---------------------------------------------------------------------------------------------
# Operator: E, B, D, B^T, E^T
# ---------------------------------------------------------------------------------------------

QF_HCURL_32 = "hcurl_32"
QF_HDIV, QF_HCURL, QF_HDIVMASS, QF_HCURLMASS, QF_H1MASS = "hdiv_33", "hcurl_33", "hdivmass_33", "hcurlmass_33", "h1_1"
QF_HCURL_22, QF_L2_1, QF_HDIVMASS_22 = "hcurl_22", "l2_1", "hdivmass_22"
QF_HDIVMASS_32, QF_HCURLMASS_22, QF_HCURLMASS_32 = "hdivmass_32", "hcurlmass_22", "hcurlmass_32"
QF_HCURL_LINE, QF_HCURLMASS_LINE = "hcurl_21|31", "hcurlmass_21|31"  # line elements: the geometry data tells 21 from 31
QF_HCURLHDIV_ERROR, QF_HDIVHCURL_ERROR = "hcurlhdiv_error_33", "hdivhcurl_error_33"
QF_HCURLHDIV, QF_HDIVHCURL = "hcurlhdiv_33", "hdivhcurl_33"  # weak curl (Interp -> Curl), mixed curl (Curl -> Interp)
QF_HCURLHDIV_22, QF_HDIVHCURL_22, QF_HDIV_22 = "hcurlhdiv_22", "hdivhcurl_22", "hdiv_22"
QF_HCURLHDIV_ERROR_22, QF_HDIVHCURL_ERROR_22 = "hcurlhdiv_error_22", "hdivhcurl_error_22"
QF_L2H1_ERROR = "l2h1_error"
QF_HDIV_32, QF_HDIV_LINE = "hdiv_32", "hdiv_21|31"  # H(div) mass on boundary / line elements
QF_L2MASS = "l2mass_*"  # div-div + H(div) mass; the geometry data tells the member (22, 33, 21, 31, 32)
QF_HCURLHDIV_32, QF_HDIVHCURL_32 = "hcurlhdiv_32", "hdivhcurl_32"
QF_HCURLHDIV_LINE, QF_HDIVHCURL_LINE = "hcurlhdiv_21|31", "hdivhcurl_21|31"
QF_HCURLH1D = "hcurlh1d_*"  # (C grad u, v) with v in a vector H1 space


class CeedOperatorOracle:
    """One libCEED sub-operator on one element block (one geometry x one integrator).

    offsets [NE, P] int32, orients [NE, P] bool or None (CeedElemRestrictionCreateOriented),
    interp / deriv dense tables, geom [NE, 11, Q], a QFunction name and its context(s).
    """

    def __init__(self, lsize, offsets, orients, interp, deriv, geom, qf, ctx, ctx2=None, vector_fe=True,
                 curl_orients=None, qw=None, deriv_comps=None):
        self.qw = qw  # quadrature weights: the extra q_w input of the 2-D curl-curl QFunctions
        self.lsize = int(lsize)
        self.off = np.asarray(offsets)
        self.NE, self.P = self.off.shape
        self.sgn = None if orients is None else np.where(np.asarray(orients), -1.0, 1.0)
        # CeedElemRestrictionCreateCurlOriented (restriction.cpp:299-369): int8 [NE, P, 3] rows
        # {sub, main, super} of the element's tridiagonal dof transformation T_e
        self.cor = None if curl_orients is None else np.asarray(curl_orients, dtype=np.float64).reshape(self.NE, self.P, 3)
        assert self.sgn is None or self.cor is None
        self.Q = geom.shape[2]
        self.vector_fe = vector_fe
        # 6 rows: 2-D; 8 rows: boundary elements (2 in 3); 4 / 5 rows: line elements in the plane / in space
        dim = 2 if geom.shape[1] in (6, 8) else (1 if geom.shape[1] in (4, 5) else 3)
        if vector_fe:
            self.interp = np.asarray(interp).reshape(dim, self.Q, self.P)
        else:
            self.interp = np.asarray(interp).reshape(1, self.Q, self.P)
        # deriv_comps: components of the derivative table when it is not the default (1: the divergence of an H(div) element)
        dc = deriv_comps if deriv_comps is not None else (1 if (dim == 2 and vector_fe) else dim)
        self.deriv = np.asarray(deriv).reshape(dc, self.Q, self.P)
        self.geom, self.qf, self.ctx, self.ctx2 = geom, qf, ctx, ctx2

    def _restrict(self, x, sl):
        u = x[self.off[sl]]
        if self.cor is not None:  # u_e = T_e x_e  [libCEED CeedElemRestrictionApply, NOTRANSPOSE]
            t = self.cor[sl]
            v = t[:, :, 1] * u
            v[:, 1:] += t[:, 1:, 0] * u[:, :-1]
            v[:, :-1] += t[:, :-1, 2] * u[:, 1:]
            return v
        return u if self.sgn is None else u * self.sgn[sl]

    def _restrict_t(self, ve, sl, unsigned=False):
        """Element-local part of E^T (before the scatter-add): T_e^T v_e, or the signs."""
        if self.cor is not None:
            t = np.abs(self.cor[sl]) if unsigned else self.cor[sl]
            w = t[:, :, 1] * ve
            w[:, :-1] += t[:, 1:, 0] * ve[:, 1:]
            w[:, 1:] += t[:, :-1, 2] * ve[:, :-1]
            return w
        if self.sgn is None or unsigned:
            return ve
        return ve * self.sgn[sl]

    def _qfunction(self, geom, ue):
        """B, D, B^T on element-local vectors ue [ne, P] -> ve [ne, P]."""
        qf = self.qf
        if qf in (QF_HDIV_32, QF_HDIV_LINE):  # H(div) mass on boundary / line elements (integ/vecfemass.cpp, cases 32 | 21 | 31)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            return np.einsum("dqj,edq->ej", self.interp, (apply_hdiv_32 if qf == QF_HDIV_32 else apply_hdiv_line)(self.ctx, geom, u))
        if qf == QF_L2MASS:  # div-div + mass on an H(div) space (integ/divdivmass.cpp): mass context first, then the scalar one
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            du = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, dv = apply_l2mass(self.ctx, self.ctx2, geom, self.qw, u, du)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, dv)
        if qf == QF_L2_1:      # 2-D curl-curl; div-div with the divergence table (integ/divdiv.cpp: l2_1 for one component)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            return np.einsum("dqj,edq->ej", self.deriv, apply_l2_1(self.ctx, geom, self.qw, cu))
        if qf == QF_HCURL_LINE:  # ND mass (values) or H1 diffusion (du/dxi) on a line element
            tabl = self.interp if self.vector_fe else self.deriv
            return np.einsum("dqj,edq->ej", tabl, apply_hcurl_line(self.ctx, geom, np.einsum("dqj,ej->edq", tabl, ue)))
        if qf == QF_HCURLMASS_LINE:
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, gv = apply_hcurlmass_line(self.ctx, self.ctx2, geom, u, gu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, gv)
        if qf == QF_HCURL_32 and not self.vector_fe:  # H1 diffusion on boundary elements (integ/diffusion.cpp, case 32)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            return np.einsum("dqj,edq->ej", self.deriv, apply_hcurl_32(self.ctx, geom, gu))
        if qf in (QF_HCURLMASS_22, QF_HCURLMASS_32):  # 2-D / boundary H1 diffusion + mass: scalar mass context first
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, gv = (apply_hcurlmass_22 if qf == QF_HCURLMASS_22 else apply_hcurlmass_32)(self.ctx, self.ctx2, geom, u, gu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, gv)
        if qf == QF_HDIVMASS_32:  # ND boundary curl-curl + mass (integ/curlcurlmass.cpp, case 32)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, cv = apply_hdivmass_32(self.ctx, self.ctx2, geom, self.qw, u, cu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, cv)
        if qf == QF_HCURL_32:  # boundary ND mass (surface impedance / absorbing / lumped-port terms)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            return np.einsum("dqj,edq->ej", self.interp, apply_hcurl_32(self.ctx, geom, u))
        if qf == QF_HCURL_22 and not self.vector_fe:  # 2-D H1 diffusion: hcurl Piola on grad u (integ/diffusion.cpp)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            return np.einsum("dqj,edq->ej", self.deriv, apply_hcurl_22(self.ctx, geom, gu))
        if qf == QF_HCURL_22:  # 2-D ND mass
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            return np.einsum("dqj,edq->ej", self.interp, apply_hcurl_22(self.ctx, geom, u))
        if qf == QF_HDIV_22:  # plane H(div) mass (integ/vecfemass.cpp:75-87 with an RT space)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            return np.einsum("dqj,edq->ej", self.interp, apply_hdiv_22(self.ctx, geom, u))
        if qf == QF_HDIVMASS_22:
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, cv = apply_hdivmass_22(self.ctx, self.ctx2, geom, self.qw, u, cu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, cv)
        if qf == QF_HCURLHDIV:  # (C u, curl v): trial Interp, test Curl (integ/mixedveccurl.cpp:75-120, same H(curl) space)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            return np.einsum("dqj,edq->ej", self.deriv, apply_hcurlhdiv_33(self.ctx, geom, u))
        if qf == QF_HDIVHCURL:  # (C curl u, v): trial Curl, test Interp (integ/mixedveccurl.cpp:21-73)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            return np.einsum("dqj,edq->ej", self.interp, apply_hdivhcurl_33(self.ctx, geom, cu))
        if qf == QF_HDIV:      # curl-curl (integ/curlcurl.cpp:48-52,60-61)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            cv = apply_hdiv_33(self.ctx, geom, cu)
            return np.einsum("dqj,edq->ej", self.deriv, cv)
        if qf == QF_HCURL and self.vector_fe:   # ND mass (integ/vecfemass.cpp)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            v = apply_hcurl_33(self.ctx, geom, u)
            return np.einsum("dqj,edq->ej", self.interp, v)
        if qf == QF_HCURL:     # H1 diffusion: hcurl Piola on grad u (integ/diffusion.cpp)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            gv = apply_hcurl_33(self.ctx, geom, gu)
            return np.einsum("dqj,edq->ej", self.deriv, gv)
        if qf == QF_HDIVMASS:  # curl-curl + mass (integ/curlcurlmass.cpp:41-45,55-64)
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            cu = np.einsum("dqj,ej->edq", self.deriv, ue)
            v, cv = apply_hdivmass_33(self.ctx, self.ctx2, geom, u, cu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, cv)
        if qf == QF_HCURLMASS:  # H1 diffusion + mass (hcurlmass_33_qf.h): mass ctx first
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            gu = np.einsum("dqj,ej->edq", self.deriv, ue)
            attr = geom[:, 0, :].astype(np.int32)
            c = self.ctx.unpack3(attr)[..., 0]  # scalar mass coefficient (1x1)
            v = (geom[:, 1, :] * c)[:, None, :] * u
            gv = apply_hcurl_33(self.ctx2, geom, gu)
            return np.einsum("dqj,edq->ej", self.interp, v) + np.einsum("dqj,edq->ej", self.deriv, gv)
        if qf == QF_H1MASS:    # H1 mass (h1_1_qf.h): v = c wdetJ u
            u = np.einsum("dqj,ej->edq", self.interp, ue)
            attr = geom[:, 0, :].astype(np.int32)
            c = self.ctx.unpack3(attr)[..., 0]
            v = (geom[:, 1, :] * c)[:, None, :] * u
            return np.einsum("dqj,edq->ej", self.interp, v)
        raise ValueError(qf)

    def apply_add(self, x, y, chunk=2048):
        """CeedOperatorApplyAdd."""
        for a in range(0, self.NE, chunk):
            sl = slice(a, min(self.NE, a + chunk))
            ve = self._restrict_t(self._qfunction(self.geom[sl], self._restrict(x, sl)), sl)
            np.add.at(y, self.off[sl].ravel(), ve.ravel())
        return y

    def element_matrices(self, sl=slice(None)):
        """A_e [ne, P, P] in native local order including orientation signs."""
        ne = self.geom[sl].shape[0]
        I = np.eye(self.P)
        Ae = np.empty((ne, self.P, self.P))
        for j in range(self.P):
            Ae[:, :, j] = self._qfunction(self.geom[sl], np.broadcast_to(I[j], (ne, self.P)))
        if self.sgn is not None:
            s = self.sgn[sl]
            Ae = Ae * s[:, :, None] * s[:, None, :]
        if self.cor is not None:  # T_e^T A_e T_e
            t = self.cor[sl]
            T = np.zeros((ne, self.P, self.P))
            r = np.arange(self.P)
            T[:, r, r] = t[:, :, 1]
            T[:, r[1:], r[:-1]] = t[:, 1:, 0]
            T[:, r[:-1], r[1:]] = t[:, :-1, 2]
            Ae = np.einsum("eai,eab,ebj->eij", T, Ae, T)
        return Ae

    def raw_element_diagonals(self, sl=slice(None)):
        """diag(B^T D B) per element, before any restriction."""
        ne = self.geom[sl].shape[0]
        I = np.eye(self.P)
        d = np.empty((ne, self.P))
        for j in range(self.P):
            d[:, j] = self._qfunction(self.geom[sl], np.broadcast_to(I[j], (ne, self.P)))[:, j]
        return d

    def assemble_sparse(self):
        import scipy.sparse as sp

        rows, cols, vals = [], [], []
        for a in range(0, self.NE, 256):
            sl = slice(a, min(self.NE, a + 256))
            Ae = self.element_matrices(sl)
            off = self.off[sl]
            rows.append(np.repeat(off[:, :, None], self.P, axis=2).ravel())
            cols.append(np.repeat(off[:, None, :], self.P, axis=1).ravel())
            vals.append(Ae.ravel())
        A = sp.coo_matrix(
            (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
            shape=(self.lsize, self.lsize),
        )
        return A.tocsr()

    def diagonal(self):
        """True diagonal of the assembled operator (libCEED's LinearAssembleAddDiagonal is exact
        for oriented restrictions; reference operator.cpp:116-143)."""
        d = np.zeros(self.lsize)
        for a in range(0, self.NE, 256):
            sl = slice(a, min(self.NE, a + 256))
            if self.cor is not None:
                # libCEED pushes the element diagonals through the transpose of the UNSIGNED copy of the
                # restriction (CeedElemRestrictionCreateUnsignedCopy in CeedOperatorLinearAssembleAddDiagonal
                # [external, published behaviour]): |T_e|^T diag(B^T D B), not diag(T_e^T A_e T_e)
                de = self._restrict_t(self.raw_element_diagonals(sl), sl, unsigned=True)
            else:
                de = np.einsum("ejj->ej", self.element_matrices(sl))
            np.add.at(d, self.off[sl].ravel(), de.ravel())
        return d


# ---------------------------------------------------------------------------------------------
# ParOperator boundary-condition logic (serial: P = identity)
# ---------------------------------------------------------------------------------------------

DIAG_ONE, DIAG_ZERO = 1, 0



class MixedSpaceOracle:
    """Two spaces on one element block: the mixed mass operator (v, C u) of BilinearForm(trial, test) +
    VectorFEMassIntegrator (fem/integ/vecfemass.cpp:88-101: f_apply_hcurlhdiv_33 for an H(curl) trial and an H(div) test
    space, f_apply_hdivhcurl_33 the other way round) and the element error integrator of
    AssembleCeedElementErrorIntegrator (fem/libceed/integrator.cpp:550-626).  `first` / `second` are CeedOperatorOracle
    objects used for their restrictions and value tables only (trial / test, or input 1 / input 2).  `first_tab` /
    `second_tab`: the [dim, Q, P] table each side enters with -- by default its value table (Interp); the gradient table
    (Grad) of an H1 space for MixedVectorGradientIntegrator (fem/integ/mixedvecgrad.cpp:43-76: f_apply_hcurl_33 | _22 with an
    H(curl) test space, f_apply_hcurlhdiv_33 | _22 with an H(div) one).  The _22 QFunctions go with plane geometry data."""

    def __init__(self, first, second, geom, qf, ctx, ctx2=None, first_tab=None, second_tab=None):
        self.a, self.b, self.geom, self.qf, self.ctx, self.ctx2 = first, second, geom, qf, ctx, ctx2
        self.ta = first.interp if first_tab is None else first_tab
        self.tb = second.interp if second_tab is None else second_tab
        assert first.NE == second.NE == geom.shape[0]

    def apply_add(self, x, y, chunk=2048):
        f = {QF_HCURLHDIV: apply_hcurlhdiv_33, QF_HDIVHCURL: apply_hdivhcurl_33, QF_HCURL: apply_hcurl_33,
             QF_HCURLHDIV_22: apply_hcurlhdiv_22, QF_HDIVHCURL_22: apply_hdivhcurl_22, QF_HCURL_22: apply_hcurl_22,
             QF_HCURLHDIV_32: apply_hcurlhdiv_32, QF_HDIVHCURL_32: apply_hdivhcurl_32, QF_HCURL_32: apply_hcurl_32,
             QF_HCURLHDIV_LINE: apply_hcurlhdiv_line, QF_HDIVHCURL_LINE: apply_hdivhcurl_line, QF_HCURL_LINE: apply_hcurl_line,
             QF_HDIV: apply_hdiv_33,  # curls of an H(curl) side (pass its curl table) and values of an H(div) one: mixedveccurl.cpp:41-46
             QF_H1MASS: apply_h1_1}[self.qf]  # QF_H1MASS: MassIntegrator between two scalar spaces (dim-1 context)
        for s0 in range(0, self.a.NE, chunk):
            sl = slice(s0, min(self.a.NE, s0 + chunk))
            u = np.einsum("dqj,ej->edq", self.ta, self.a._restrict(x, sl))
            ve = np.einsum("dqj,edq->ej", self.tb, f(self.ctx, self.geom[sl], u))
            np.add.at(y, self.b.off[sl].ravel(), self.b._restrict_t(ve, sl).ravel())
        return y

    def gradient_add(self, x, y, comp_stride, chunk=2048):
        """GradientIntegrator (integ/grad.cpp:16-72, f_apply_hcurlh1d_*): `first` is the scalar H1 trial space (its gradient
        table passed as `first_tab`), `second` the scalar space whose value table every component of the vector H1 test
        space uses; component c of test dof j lives at y[c * comp_stride + j] (restriction.cpp:137-142, byNODES)."""
        for s0 in range(0, self.a.NE, chunk):
            sl = slice(s0, min(self.a.NE, s0 + chunk))
            gu = np.einsum("dqj,ej->edq", self.ta, self.a._restrict(x, sl))
            v = apply_hcurlh1d(self.ctx, self.geom[sl], gu)  # [ne, sdim, Q]
            for c in range(v.shape[1]):
                ve = np.einsum("qj,eq->ej", self.tb[0], v[:, c, :])
                np.add.at(y, (c * comp_stride + self.b.off[sl]).ravel(), self.b._restrict_t(ve, sl).ravel())
        return y

    def error_add(self, u1, u2, est, chunk=2048):
        """est[e] += sum_q of the error QFunction (the all-ones `mesh_elem_basis`, integrator.cpp:560-574)."""
        f = {QF_HCURLHDIV_ERROR: apply_hcurlhdiv_error_33, QF_HDIVHCURL_ERROR: apply_hdivhcurl_error_33,
             QF_HCURLHDIV_ERROR_22: apply_hcurlhdiv_error_22, QF_HDIVHCURL_ERROR_22: apply_hdivhcurl_error_22,
             QF_L2H1_ERROR: apply_l2h1_error}[self.qf]
        for s0 in range(0, self.a.NE, chunk):
            sl = slice(s0, min(self.a.NE, s0 + chunk))
            q1 = np.einsum("dqj,ej->edq", self.ta, self.a._restrict(u1, sl))
            q2 = np.einsum("dqj,ej->edq", self.tb, self.b._restrict(u2, sl))
            est[sl] += f(self.ctx, self.ctx2, self.geom[sl], q1, q2).sum(axis=1)
        return est

class ParOperatorOracle:
    """rap.cpp:195-234 with a trivial prolongation (one rank)."""

    def __init__(self, ops, ess, diag_policy=DIAG_ONE):
        self.ops = list(ops)
        self.ess = np.asarray(ess, dtype=np.int64)
        self.policy = diag_policy
        self.n = self.ops[0].lsize

    def mult(self, x):
        tx = x.copy()
        tx[self.ess] = 0.0
        y = np.zeros(self.n)
        for op in self.ops:
            op.apply_add(tx, y)
        y[self.ess] = x[self.ess] if self.policy == DIAG_ONE else 0.0
        return y

    def add_mult(self, x, y, a=1.0):
        """rap.cpp:277-318."""
        return y + a * self.mult(x)

    def eliminate_rhs(self, x, b):
        """rap.cpp:56-82."""
        tx = np.zeros(self.n)
        tx[self.ess] = x[self.ess]
        y = np.zeros(self.n)
        for op in self.ops:
            op.apply_add(tx, y)
        b = b - y
        b[self.ess] = x[self.ess] if self.policy == DIAG_ONE else 0.0
        return b

    def diagonal(self):
        d = np.zeros(self.n)
        for op in self.ops:
            d += op.diagonal()
        d[self.ess] = 1.0 if self.policy == DIAG_ONE else 0.0
        return d


# ---------------------------------------------------------------------------------------------
# Krylov loop
# ---------------------------------------------------------------------------------------------

def spectral_norm_power(mult, n, tol=1e-4, max_it=1000, seed=0, u0=None):
    """linalg/operator.cpp:583-631 with herm=True."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, n) if u0 is None else u0.copy()
    u /= np.linalg.norm(u)
    l0 = 0.0
    l = 0.0
    it = 0
    while it < max_it:
        v = mult(u)
        u = v
        l = np.linalg.norm(u)
        u = u / l
        if it > 0 and abs(l - l0) / l0 < tol:
            break
        l0 = l
        it += 1
    return l


class ChebyshevOracle:
    """chebyshev.cpp:160-220 (4th kind, default) and :222-293 (1st kind, `first_kind=True`; sf_min <= 0 takes the
    optimised lambda_min estimate of Phillips and Fischer, chebyshev.cpp:244-247)."""

    def __init__(self, A, order, sf_max=1.0, lambda_max=None, u0=None, first_kind=False, sf_min=0.0, smooth_it=1):
        self.A, self.order, self.first_kind, self.pc_it = A, order, first_kind, smooth_it
        self.dinv = 1.0 / A.diagonal()
        if lambda_max is None:
            lambda_max = spectral_norm_power(lambda u: self.dinv * A.mult(u), A.n, u0=u0)
        self.lambda_max = sf_max * lambda_max
        if first_kind:
            if sf_min <= 0.0:
                sf_min = 1.69 / (order ** 1.68 + 2.11 * order + 1.98)
            lambda_min = sf_min * self.lambda_max
            self.theta = 0.5 * (self.lambda_max + lambda_min)
            self.delta = 0.5 * (self.lambda_max - lambda_min)

    def mult2(self, x, y, initial_guess):
        for it in range(self.pc_it):
            if initial_guess or it > 0:
                r = x - self.A.mult(y)
            else:
                r = x.copy()
                y = np.zeros_like(x)
            if not self.first_kind:
                lam = self.lambda_max
                d = 4.0 / (3.0 * lam) * self.dinv * r
                for k in range(1, self.order):
                    y = y + d
                    r = r - self.A.mult(d)
                    sd = (2.0 * k - 1.0) / (2.0 * k + 3.0)
                    sr = (8.0 * k + 4.0) / ((2.0 * k + 3.0) * lam)
                    d = sd * d + sr * self.dinv * r
            else:
                d = (1.0 / self.theta) * self.dinv * r
                rhop = self.delta / self.theta
                for k in range(1, self.order):
                    y = y + d
                    r = r - self.A.mult(d)
                    rho = 1.0 / (2.0 * self.theta / self.delta - rhop)
                    d = (rho * rhop) * d + (2.0 * rho / self.delta) * self.dinv * r
                    rhop = rho
            y = y + d
        return y

    mult_transpose2 = mult2  # chebyshev.hpp: MultTranspose2 forwards to Mult2 (the polynomial in D^-1 A is symmetric)


class DistRelaxationOracle:
    """linalg/distrelaxation.cpp:98-151: Hiptmair distributive relaxation.  A: ParOperatorOracle-like on the Nedelec
    space, A_G on the auxiliary H1 space, G = (mult, mult_transpose) discrete gradient, B / B_G Chebyshev oracles
    (cheby_smooth_it = 1 as gmg.cpp:41-47 configures them), ess_G the essential true dofs of the auxiliary space."""

    def __init__(self, A, A_G, G, B, B_G, ess_G, smooth_it=1):
        self.A, self.A_G, self.G, self.B, self.B_G, self.ess_G, self.pc_it = A, A_G, G, B, B_G, np.asarray(ess_G), smooth_it

    def mult2(self, x, y, initial_guess):
        for it in range(self.pc_it):
            # y = y + B (x - A y)
            y = self.B.mult2(x, y, initial_guess or it > 0)
            # y = y + G B_G G^T (x - A y)
            r = x - self.A.mult(y)
            x_G = self.G[1](r)
            if self.ess_G.size:
                x_G[self.ess_G] = 0.0
            y_G = self.B_G.mult2(x_G, None, False)
            y = y + self.G[0](y_G)
        return y

    def mult_transpose2(self, x, y, initial_guess):
        for it in range(self.pc_it):
            # y = y + G B_G^T G^T (x - A y)
            if initial_guess or it > 0:
                r = x - self.A.mult(y)
                x_G = self.G[1](r)
            else:
                y = np.zeros_like(x)
                x_G = self.G[1](x)
            if self.ess_G.size:
                x_G[self.ess_G] = 0.0
            y_G = self.B_G.mult_transpose2(x_G, None, False)
            y = y + self.G[0](y_G)
            # y = y + B^T (x - A y)
            y = self.B.mult_transpose2(x, y, True)
        return y


class GMGOracle:
    """gmg.cpp:171-205.  A[l] ParOperatorOracle per level (0 = coarsest), P[l] prolongation
    callables (mult, mult_transpose) from level l to l+1, B[l] smoothers, B[0] coarse solver
    callable x -> y."""

    def __init__(self, A, P, smoothers, coarse_solve, ess_lists):
        self.A, self.P, self.B, self.coarse, self.ess = A, P, smoothers, coarse_solve, ess_lists

    def vcycle(self, l, x, y, initial_guess):
        if l == 0:
            return self.coarse(x)
        y = self.B[l].mult2(x, y, initial_guess)
        r = x - self.A[l].mult(y)
        xc = self.P[l - 1][1](r)
        xc[self.ess[l - 1]] = 0.0
        yc = self.vcycle(l - 1, xc, None, False)
        y = y + self.P[l - 1][0](yc)
        return self.B[l].mult_transpose2(x, y, True)  # gmg.cpp:202-204

    def mult(self, x):
        return self.vcycle(len(self.A) - 1, x, None, False)


def pcg(A_mult, b, B_mult=None, rel_tol=0.0, abs_tol=0.0, max_it=100):
    """iterative.cpp:360-486, zero initial guess.  Returns (x, iterations, residual history)."""
    x = np.zeros_like(b)
    r = b.copy()
    z = B_mult(r) if B_mult else r.copy()
    beta = float(z @ r)
    res = np.sqrt(abs(beta))
    initial_res = res
    eps = max(rel_tol * initial_res, abs_tol)
    hist = [res]
    it = 0
    converged = res < eps
    beta_prev = 0.0
    p = None
    while it < max_it and not converged:
        p = z.copy() if it == 0 else z + (beta / beta_prev) * p
        z = A_mult(p)
        denom = float(z @ p)
        alpha = beta / denom
        x = x + alpha * p
        r = r - alpha * z
        beta_prev = beta
        z = B_mult(r) if B_mult else r.copy()
        beta = float(z @ r)
        res = np.sqrt(abs(beta))
        hist.append(res)
        converged = res < eps
        it += 1
    return x, it, hist


def _generate_plane_rotation(dx, dy):
    """iterative.cpp:72-166 (d/zlartg; the unscaled branches -- inputs of a Krylov recursion are far from over/underflow)."""
    if np.iscomplexobj(dx) or np.iscomplexobj(dy):
        dx, dy = complex(dx), complex(dy)
        if dy == 0:
            return 1.0, 0j
        if dx == 0:
            return 0.0, np.conj(dy) / abs(dy)
        dx2, dy2 = abs(dx) ** 2, abs(dy) ** 2
        dz2 = dx2 + dy2
        return np.sqrt(dx2 / dz2), np.conj(dy) * (dx / np.sqrt(dx2 * dz2))
    if dy == 0.0:
        return 1.0, 0.0
    if dx == 0.0:
        return 0.0, np.copysign(1.0, dy)
    d = np.sqrt(dx * dx + dy * dy)
    return abs(dx) / d, dy / np.copysign(d, dx)


def gmres(A_mult, b, B_mult=None, rel_tol=0.0, abs_tol=0.0, max_it=100, max_dim=-1, pc_side="left", flexible=False,
          x0=None):
    """GmresSolver::Mult (iterative.cpp:543-705) and FgmresSolver::Mult (:733-871), MGS (orthog.hpp:41-55), real or
    complex (Dot(x, y) = y^H x, vector.cpp:674-685).  Returns (x, iterations, residual history, converged)."""
    cplx = np.iscomplexobj(b)
    dtype = complex if cplx else float
    n = b.size
    if max_dim < 0:
        max_dim = max_it
    if flexible:
        assert B_mult is not None
        pc_side = "right"
    left = B_mult is not None and pc_side == "left"
    right = B_mult is not None and pc_side == "right"
    initial_guess = x0 is not None
    x = x0.astype(dtype).copy() if initial_guess else np.zeros(n, dtype=dtype)
    V = [None] * (max_dim + 1)
    Z = [None] * (max_dim + 1)
    H = np.zeros((max_dim + 1, max_dim), dtype=dtype)
    s = np.zeros(max_dim + 1, dtype=dtype)
    cs = np.zeros(max_dim + 1)
    sn = np.zeros(max_dim + 1, dtype=dtype)
    hist, it, restart, beta, eps, converged = [], 0, 0, 0.0, 0.0, False
    while it < max_it:
        guess = initial_guess or restart > 0
        if left:
            r = B_mult(b - A_mult(x)) if guess else B_mult(b)
        else:
            r = b - A_mult(x) if guess else b.astype(dtype).copy()
        if not guess:
            x[:] = 0
        true_beta = np.linalg.norm(r)
        if it == 0:
            if initial_guess:
                initial_res = np.linalg.norm(B_mult(b)) if left else np.linalg.norm(b)
            else:
                initial_res = true_beta
            eps = max(rel_tol * initial_res, abs_tol)
        beta = true_beta
        if beta < eps:
            converged = True
            break
        V[0] = r / beta
        s[:] = 0
        s[0] = beta
        j = 0
        while True:
            hist.append(beta)
            if left:
                w = B_mult(A_mult(V[j]))
            elif right:
                Z[j] = B_mult(V[j])
                w = A_mult(Z[j])
            else:
                w = A_mult(V[j])
            for i in range(j + 1):
                H[i, j] = np.vdot(V[i], w)
                w = w - H[i, j] * V[i]
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = w / H[j + 1, j]
            for k in range(j):
                t = cs[k] * H[k, j] + sn[k] * H[k + 1, j]
                H[k + 1, j] = -np.conj(sn[k]) * H[k, j] + cs[k] * H[k + 1, j]
                H[k, j] = t
            c, sgn = _generate_plane_rotation(H[j, j], H[j + 1, j])
            cs[j], sn[j] = c, sgn
            t = cs[j] * H[j, j] + sn[j] * H[j + 1, j]
            H[j + 1, j] = -np.conj(sn[j]) * H[j, j] + cs[j] * H[j + 1, j]
            H[j, j] = t
            t = cs[j] * s[j] + sn[j] * s[j + 1]
            s[j + 1] = -np.conj(sn[j]) * s[j] + cs[j] * s[j + 1]
            s[j] = t
            beta = abs(s[j + 1])
            converged = beta < eps
            if converged or j + 1 == max_dim or it + 1 == max_it:
                it += 1
                break
            j += 1
            it += 1
        for i in range(j, -1, -1):
            s[i] /= H[i, i]
            for k in range(i - 1, -1, -1):
                s[k] -= H[k, i] * s[i]
        if flexible:
            for k in range(j + 1):
                x = x + s[k] * Z[k]
        elif not right:
            for k in range(j + 1):
                x = x + s[k] * V[k]
        else:
            rr = np.zeros(n, dtype=dtype)
            for k in range(j + 1):
                rr = rr + s[k] * V[k]
            x = x + B_mult(rr)
        if converged:
            break
        restart += 1
    hist.append(beta)
    return x, it, hist, converged


# ---------------------------------------------------------------------------------------------
# p-prolongation (fem/bilinearform.cpp:203-282, fem/libceed/basis.cpp:116-165): element matrix =
# nodal interpolation of the coarse shape functions at the fine dofs (what MFEM's
# GetTransferMatrix returns for nodal elements), applied through E / E^T with the inverse
# multiplicity of the fine restriction.
# ---------------------------------------------------------------------------------------------

def nd_hex_interp_lex(pc, pf):
    """Dense [P_f, P_c] element interpolation matrix in tensor (lexicographic) dof order."""
    cpc, opc = gll_points(pc + 1), gl_points(pc)[0]
    cpf, opf = gll_points(pf + 1), gl_points(pf)[0]
    Pc, Pf = 3 * pc * (pc + 1) ** 2, 3 * pf * (pf + 1) ** 2
    M = np.zeros((Pf, Pc))
    for comp in range(3):
        ncd = [pc + 1] * 3
        ncd[comp] = pc
        nfd = [pf + 1] * 3
        nfd[comp] = pf
        nodes_c = [cpc] * 3
        nodes_c[comp] = opc
        nodes_f = [cpf] * 3
        nodes_f[comp] = opf
        for kf in range(nfd[2]):
            for jf in range(nfd[1]):
                for i_f in range(nfd[0]):
                    lf = comp * pf * (pf + 1) ** 2 + i_f + nfd[0] * (jf + nfd[1] * kf)
                    pt = (nodes_f[0][i_f], nodes_f[1][jf], nodes_f[2][kf])
                    for kc in range(ncd[2]):
                        for jc in range(ncd[1]):
                            for ic in range(ncd[0]):
                                lc = comp * pc * (pc + 1) ** 2 + ic + ncd[0] * (jc + ncd[1] * kc)
                                M[lf, lc] = (lagrange(nodes_c[0], pt[0], ic)[0] * lagrange(nodes_c[1], pt[1], jc)[0]
                                             * lagrange(nodes_c[2], pt[2], kc)[0])
    return M


class InterpOracle:
    """y_f = D^-1 sum_e E_f^T I E_c x_c and its transpose (serial)."""

    def __init__(self, dof_c, sgn_c, dof_f, sgn_f, n_c, n_f, M):
        self.dc, self.sc, self.df, self.sf = dof_c, sgn_c.astype(np.float64), dof_f, sgn_f.astype(np.float64)
        self.nc, self.nf, self.M = n_c, n_f, M
        mult = np.zeros(n_f)
        np.add.at(mult, dof_f.ravel(), 1.0)
        self.inv_mult = 1.0 / mult

    def mult(self, x):
        ue = x[self.dc] * self.sc
        ve = (ue @ self.M.T) * self.sf
        y = np.zeros(self.nf)
        np.add.at(y, self.df.ravel(), ve.ravel())
        return y * self.inv_mult

    def mult_transpose(self, x):
        ue = (x * self.inv_mult)[self.df] * self.sf
        ve = (ue @ self.M) * self.sc
        y = np.zeros(self.nc)
        np.add.at(y, self.dc.ravel(), ve.ravel())
        return y


# ---------------------------------------------------------------------------------------------
# h-refinement transfer: the prolongation between the spaces of one collection on a mesh and on its uniform refinement,
# mfem::TransferOperator for two different meshes (reference fem/fespace.cpp:246-251, the h-levels of
# fem/multigrid.hpp:103-112).  MFEM [external, published behaviour: FiniteElementSpace::RefinementOperator built from
# Mesh::GetRefinementTransforms()] applies, element by element of the fine mesh, the local interpolation matrix
# FiniteElement::GetLocalInterpolation(T) of the child's embedding T into its parent: row k = the fine dof functional k of
# the child applied to the parent's basis.  For a covariant (H(curl)) element with the affine embedding x_parent = o + A x_child
# that functional is  phi_parent(o + A x_k) . (A t_k); for H1 the point value.  Every copy of a shared fine dof gets the same
# value (conforming spaces): MFEM keeps one, which is the D^-1 sum of equal copies InterpOracle computes.
# ---------------------------------------------------------------------------------------------

def hex_refinement_matrices(p, hcurl=True):
    """[8][P, P] local interpolation matrices of the eight children (octant a + 2 b + 4 c of the parent's reference cube, the
    order of palace_amd.fem.mesh.refine_uniform) of an order-p tensor element, tensor (lexicographic) dof order on both sides."""
    cp, op = gll_points(p + 1), (gl_points(p)[0] if hcurl else None)

    def one_d(nodes, half, scale):
        # rows: child nodes mapped into the parent, columns: parent 1-D basis functions
        return np.array([[scale * lagrange(nodes, 0.5 * (half + x), j)[0] for j in range(len(nodes))] for x in nodes])

    out = []
    for c in range(2):
        for b in range(2):
            for a in range(2):
                h = (a, b, c)
                if not hcurl:
                    M = np.kron(one_d(cp, h[2], 1.0), np.kron(one_d(cp, h[1], 1.0), one_d(cp, h[0], 1.0)))
                else:
                    blocks = []
                    for comp in range(3):
                        m1 = [one_d(op, h[d], 0.5) if d == comp else one_d(cp, h[d], 1.0) for d in range(3)]
                        blocks.append(np.kron(m1[2], np.kron(m1[1], m1[0])))
                    n = blocks[0].shape[0]
                    M = np.zeros((3 * n, 3 * n))
                    for comp in range(3):
                        M[comp * n:(comp + 1) * n, comp * n:(comp + 1) * n] = blocks[comp]
                out.append(M)
    return np.array(out)


class RefinementTransferOracle:
    """y_f = D^-1 sum_e E_f^T M[mat_id[e]] E_parent(e) x_c over the FINE elements e, and its transpose (serial).
    dof_c / sgn_c [ne_f, P_c]: dofs and signs of every fine element's PARENT in the coarse space."""

    def __init__(self, dof_c, sgn_c, dof_f, sgn_f, n_c, n_f, Ms, mat_id):
        self.dc, self.sc, self.df, self.sf = dof_c, sgn_c.astype(np.float64), dof_f, sgn_f.astype(np.float64)
        self.nc, self.nf, self.Ms, self.mid = n_c, n_f, np.asarray(Ms), np.asarray(mat_id)
        mult = np.zeros(n_f)
        np.add.at(mult, dof_f.ravel(), 1.0)
        self.inv_mult = 1.0 / mult

    def mult(self, x):
        ue = x[self.dc] * self.sc
        ve = np.einsum("eij,ej->ei", self.Ms[self.mid], ue) * self.sf
        y = np.zeros(self.nf)
        np.add.at(y, self.df.ravel(), ve.ravel())
        return y * self.inv_mult

    def mult_transpose(self, x):
        ue = (x * self.inv_mult)[self.df] * self.sf
        ve = np.einsum("eij,ei->ej", self.Ms[self.mid], ue) * self.sc
        y = np.zeros(self.nc)
        np.add.at(y, self.dc.ravel(), ve.ravel())
        return y


def nd_hex_gradient_lex(p):
    """Dense [P_ND, P_H1] discrete gradient of the order-p hex pair in tensor dof order: the ND dof
    (C; i, j, k) of grad(phi) is d/dx_C of the H1 interpolant at the ND node (nodal interpolation of
    the gradient, what mfem::GradientInterpolator / ProjectGrad builds; basis.cpp:139-143)."""
    cp, op = gll_points(p + 1), gl_points(p)[0]
    n1 = p + 1
    M = np.zeros((3 * p * n1 * n1, n1**3))
    for comp in range(3):
        nd = [n1] * 3
        nd[comp] = p
        for k in range(nd[2]):
            for j in range(nd[1]):
                for i in range(nd[0]):
                    row = comp * p * n1 * n1 + i + nd[0] * (j + nd[1] * k)
                    idx = [i, j, k]
                    for a in range(n1):
                        col_idx = list(idx)
                        col_idx[comp] = a
                        col = col_idx[0] + n1 * (col_idx[1] + n1 * col_idx[2])
                        M[row, col] = lagrange(cp, op[idx[comp]], a)[1]
    return M


class DenseInterpOracle:
    """The interpolator operator of DiscreteLinearOperator::PartialAssemble (bilinearform.cpp:203-282) for
    general elements: y = D^-1 sum_e E_range^T M E_domain x, D = dof multiplicity of the range space
    (:256-279); restrictions plain / oriented / curl-oriented, the range one in its interpolator-range
    (dual inverse) form (restriction.cpp:318-336)."""

    def __init__(self, dom, rng, M):
        self.M = np.asarray(M, dtype=np.float64)
        self.dom, self.rng = dom, rng
        self.mult_r = np.bincount(np.asarray(rng["offsets"]).ravel(), minlength=rng["lsize"]).astype(np.float64)

    @staticmethod
    def _apply_rows(r, u):
        """u_e = T u (rows of the tridiagonal / sign matrix applied)."""
        if r.get("curl_orients") is not None:
            t = np.asarray(r["curl_orients"], dtype=np.float64)
            v = t[:, :, 1] * u
            v[:, 1:] += t[:, 1:, 0] * u[:, :-1]
            v[:, :-1] += t[:, :-1, 2] * u[:, 1:]
            return v
        if r.get("orients") is not None:
            return u * np.where(np.asarray(r["orients"]), -1.0, 1.0)
        return u

    @staticmethod
    def _apply_rows_t(r, v):
        if r.get("curl_orients") is not None:
            t = np.asarray(r["curl_orients"], dtype=np.float64)
            w = t[:, :, 1] * v
            w[:, :-1] += t[:, 1:, 0] * v[:, 1:]
            w[:, 1:] += t[:, :-1, 2] * v[:, :-1]
            return w
        if r.get("orients") is not None:
            return v * np.where(np.asarray(r["orients"]), -1.0, 1.0)
        return v

    def mult(self, x):
        u = self._apply_rows(self.dom, x[self.dom["offsets"]])
        w = self._apply_rows_t(self.rng, u @ self.M.T)
        y = np.zeros(self.rng["lsize"])
        np.add.at(y, np.asarray(self.rng["offsets"]).ravel(), w.ravel())
        return y / self.mult_r

    def mult_transpose(self, x):
        z = (x / self.mult_r)[self.rng["offsets"]]
        u = self._apply_rows(self.rng, z) @ self.M
        w = self._apply_rows_t(self.dom, u)
        y = np.zeros(self.dom["lsize"])
        np.add.at(y, np.asarray(self.dom["offsets"]).ravel(), w.ravel())
        return y


# ---------------------------------------------------------------------------------------------
# Orthogonalisation of a Krylov column (linalg/orthog.hpp:41-89), as the reference's free functions:
# the inputs V[j] are assumed normalised, w is not normalised on return.
# ---------------------------------------------------------------------------------------------

def orthogonalize_column(kind, V, w, m, weight=None, global_sum=None):
    """OrthogonalizeColumnMGS (kind "MGS", :41-55) / OrthogonalizeColumnCGS (kind "CGS", "CGS2" = refine, :57-89).
    Returns (H, w_new).  Inner product dot_op(w, v) = v^H (W w) (LocalDot(x, y) = y^H x, vector.cpp:674-685; the
    weighted helper of test/unit/test-orthog.cpp:21-68 applies the real W to w first).  global_sum(array) -> array is
    Mpi::GlobalSum over the communicator (identity on one rank): one call per coefficient for MGS, one per pass for
    CGS."""
    gs = (lambda a: a) if global_sum is None else global_sum
    w = np.array(w, dtype=np.result_type(w, *[v for v in V[:m]]) if m else None, copy=True)
    H = np.zeros(m, dtype=w.dtype)

    def dot(x, y):
        wx = x if weight is None else weight @ x
        return np.vdot(y, wx)

    if kind == "MGS":
        for j in range(m):
            H[j] = gs(np.array([dot(w, V[j])]))[0]
            w = w - H[j] * V[j]
        return H, w
    if m == 0:
        return H, w
    for j in range(m):
        H[j] = dot(w, V[j])
    H[:] = gs(H)
    for j in range(m):
        w = w - H[j] * V[j]
    if kind == "CGS2":
        dH = gs(np.array([dot(w, V[j]) for j in range(m)]))
        for j in range(m):
            H[j] += dH[j]
            w = w - dH[j] * V[j]
    return H, w


# ---------------------------------------------------------------------------------------------
# Mixed (trial space != test space) operators: fem/integ/mixedveccurl.cpp
# ---------------------------------------------------------------------------------------------

class MixedCurlOperatorOracle:
    """MixedVectorCurlIntegrator with an H(div) test space (mixedveccurl.cpp:22-68): (Q curl u, v) for u in ND, v in RT,
    y = E_test^T B_test^T D G_trial E_trial x with D = f_apply_hdiv_33 (both factors are H(div)-mapped), trial ops Curl,
    test ops Interp.  MixedVectorWeakCurlIntegrator with an H(div) trial space (:70-117) is its transpose with the
    coefficient scaled by -1: `weak=True` applies -(this)^T.
    trial / test: dicts with offsets [NE, P], lsize and orients | curl_orients (native restrictions)."""

    def __init__(self, trial, test, trial_curl, test_interp, geom, ctx):
        self.tr = CeedOperatorOracle(trial["lsize"], trial["offsets"], trial.get("orients"), trial_curl, trial_curl, geom,
                                     QF_HDIV, ctx, curl_orients=trial.get("curl_orients"))
        self.te = CeedOperatorOracle(test["lsize"], test["offsets"], test.get("orients"), test_interp, test_interp, geom,
                                     QF_HDIV, ctx, curl_orients=test.get("curl_orients"))
        self.geom, self.ctx = geom, ctx

    def mult(self, x, chunk=2048, weak=False):
        a, b = (self.te, self.tr) if weak else (self.tr, self.te)  # a: input side, b: output side
        y = np.zeros(b.lsize)
        for s in range(0, a.NE, chunk):
            sl = slice(s, min(a.NE, s + chunk))
            ue = a._restrict(x, sl)
            cu = np.einsum("dqj,ej->edq", a.deriv, ue)      # curl (trial) or value (weak: RT values) at the points
            cv = apply_hdiv_33(self.ctx, self.geom[sl], cu)  # symmetric coefficients: D^T = D
            ve = b._restrict_t(np.einsum("dqj,edq->ej", b.deriv, cv), sl)
            np.add.at(y, b.off[sl].ravel(), ve.ravel())
        return -y if weak else y


# ---- native coarse solvers: the cycles restated on host matrices (checker for palace_amd/csrc/amg_solver.hip) ---------------
# The reference calls HYPRE here (linalg/amg.cpp:12-49, linalg/ams.cpp:18-224); HYPRE is not in /root/reference, so there is no
# reference output to pin these on: "parity unpinned" for the coarse solvers themselves.  What IS checked: the device cycle
# equals this restatement on the very hierarchy the library built (tests/test_ams_gpu.py, 1e-10), the preconditioners are
# symmetric positive definite, and the preconditioned solves reproduce sparse direct solutions.

def cheb4_l1(A, b, x, order, zero_guess):
    """4th-kind Chebyshev smoothing (chebyshev.cpp:190-220) on D_l1^-1 A with lambda_max = 1, D_l1 = diag(sum_j |a_ij|)."""
    l1 = np.asarray(abs(A).sum(axis=1)).ravel()
    dinv = np.where(l1 > 0.0, 1.0 / np.where(l1 > 0.0, l1, 1.0), 0.0)
    if zero_guess:
        r = b.copy()
        x = np.zeros_like(b)
    else:
        r = b - A @ x
    d = (4.0 / 3.0) * dinv * r
    for k in range(1, order):
        x = x + d
        r = r - A @ d
        d = (2.0 * k - 1.0) / (2.0 * k + 3.0) * d + (8.0 * k + 4.0) / (2.0 * k + 3.0) * dinv * r
    return x + d


def amg_vcycle(A, P, cinv, b, order=2, level=0):
    """One V-cycle from a zero guess over the hierarchy (A[l], P[l]) with the dense inverse `cinv` on the last level."""
    if level + 1 == len(A):
        return cinv @ b
    x = cheb4_l1(A[level], b, None, order, True)
    r = b - A[level] @ x
    xc = amg_vcycle(A, P, cinv, P[level].T @ r, order, level + 1)
    x = x + P[level] @ xc
    return cheb4_l1(A[level], b, x, order, False)


def ams_cycle(A, G, Pi, amg_G, amg_Pi, b, order=2, singular=False):
    """HYPRE AMS cycle type 14 (ams.cpp:24-28: 0 1 (3 + 4 + 5) 1 0) from a zero guess: smoothing on A, gradient-space
    correction, additive scalar nodal-space corrections (one block-diagonal solve), gradient space, smoothing.
    amg_G / amg_Pi: callables applying the auxiliary-space solves."""
    x = cheb4_l1(A, b, None, order, True)

    def correct(T, B, x):
        return x + T @ B(T.T @ (b - A @ x))

    if not singular:
        x = correct(G, amg_G, x)
    x = correct(Pi, amg_Pi, x)
    if not singular:
        x = correct(G, amg_G, x)
    return cheb4_l1(A, b, x, order, False)


def ams_nodal_interpolation(G, coords, ess_flag):
    """Pi = [Pi_x Pi_y Pi_z], Pi_c = |G| diag(G x_c) / 2 (HYPRE_AMSSetCoordinateVectors), rows of essential edges dropped;
    returns (G without those rows, Pi)."""
    import scipy.sparse as sp

    keep = sp.diags((~np.asarray(ess_flag, dtype=bool)).astype(np.float64))
    Gb = (keep @ G).tocsr()
    blocks = [(sp.diags(0.5 * (G @ coords[:, c])) @ abs(Gb)).tocsr() for c in range(coords.shape[1])]
    return Gb, sp.hstack(blocks).tocsr()
