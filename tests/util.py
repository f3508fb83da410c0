"""Shared test helpers: build the oracle's view of a problem from the same descriptors the HIP
library receives (native-ordered oriented restriction, dense tables, geometry data)."""
import numpy as np

from oracle import capi
from oracle import palace_oracle as po
from palace_amd.fem.fespace import NDHexSpace

_dense_cache = {}


def dense_tables(nd: NDHexSpace, q1d):
    key = (nd.p, q1d)
    if key not in _dense_cache:
        _dense_cache[key] = po.nd_hex_dense_tables(nd.p, q1d, nd.dof_map_native())
    return _dense_cache[key]


def oracle_geom(mesh, q1d):
    """geom [NE, 11, Q] through the oracle (mesh-node grad table + geom_33 restatement)."""
    _, wts = po.hex_quadrature(q1d)
    G = po.mesh_q2_grad_table(q1d)
    J = np.einsum("dqn,eni->eqid", G, mesh.elem_coords())  # J[e,q,i,d]
    Jcm = np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9)
    return po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, Jcm)


def make_ctx(kind, nattr=1):
    """Coefficient contexts used across tests: returns (oracle CoeffCtx, raw blob)."""
    if kind == "identity":
        c = po.CoeffCtx()
    elif kind == "scalar":
        c = po.CoeffCtx(attr_mat=[0] * nattr, mat_coeff=[np.array([2.08])])
    elif kind == "aniso":
        rng = np.random.default_rng(7)
        A = rng.uniform(-1, 1, (3, 3))
        spd = A @ A.T + 2.0 * np.eye(3)
        mats = [spd, np.array([0.7])]
        c = po.CoeffCtx(attr_mat=[i % 2 for i in range(nattr)], mat_coeff=mats, a=1.3)
    elif kind == "nonsym":  # general (non-symmetric) 3x3 material: only the matrix-free D can apply it
        rng = np.random.default_rng(9)
        A = rng.uniform(-1, 1, (3, 3)) + 3.0 * np.eye(3)
        c = po.CoeffCtx(attr_mat=[i % 2 for i in range(nattr)], mat_coeff=[A, np.array([0.7])], a=1.1)
    else:
        raise ValueError(kind)
    return c, c.pack()


QF_MAP = {"hcurlhdiv": (po.QF_HCURLHDIV, None), "hdivhcurl": (po.QF_HDIVHCURL, None),  # numpy oracle only
          "hdiv": (po.QF_HDIV, capi.QF_HDIV), "hcurl": (po.QF_HCURL, capi.QF_HCURL),
          "hdivmass": (po.QF_HDIVMASS, capi.QF_HDIVMASS)}


def oracle_apply_c(nd, geom, qf, blob, x, q1d):
    """y = A x through the C oracle (dense tables, oriented restriction)."""
    off, ori = nd.native_restriction()
    interp, curl = dense_tables(nd, q1d)
    y = np.zeros(nd.ndofs)
    capi.apply_add(off, ori, interp, curl, geom, QF_MAP[qf][1], blob, np.ascontiguousarray(x), y)
    return y


def oracle_operator(nd, geom, qf, ctx, ctx2=None, q1d=None):
    off, ori = nd.native_restriction()
    interp, curl = dense_tables(nd, q1d)
    return po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, geom, QF_MAP[qf][0], ctx, ctx2)


class FastParOperatorOracle(po.ParOperatorOracle):
    """ParOperatorOracle whose local apply runs through the C oracle (same restatement, compiled)."""

    def __init__(self, nd, geom, qf, blob, ess, q1d, ctx, ctx2=None, policy=po.DIAG_ONE):
        self.nd, self.geom, self.qf, self.blob, self.q1d = nd, geom, qf, blob, q1d
        self.ess = np.asarray(ess, dtype=np.int64)
        self.policy = policy
        self.n = nd.ndofs
        self._np_op = oracle_operator(nd, geom, qf, ctx, ctx2, q1d)
        self._diag = None

    def mult(self, x):
        tx = x.copy()
        tx[self.ess] = 0.0
        y = oracle_apply_c(self.nd, self.geom, self.qf, self.blob, tx, self.q1d)
        y[self.ess] = x[self.ess] if self.policy == po.DIAG_ONE else 0.0
        return y

    def diagonal(self):
        if self._diag is None:
            d = self._np_op.diagonal()
            d[self.ess] = 1.0 if self.policy == po.DIAG_ONE else 0.0
            self._diag = d
        return self._diag


def nd_interpolate(space, F):
    """Nodal interpolant of a smooth vector field F(x) -> [.., 3] in an NDHexSpace-like space:
    dof = F(x_node) . (J e_c) (covariant Piola), written through the signed element->dof map.
    Returns the local vector (size space.ndofs)."""
    from palace_amd.fem.basis1d import gauss_legendre, gauss_lobatto
    from palace_amd.fem.mesh import _q2_1d

    p, mesh = space.p, space.mesh
    cp, op = gauss_lobatto(p + 1), gauss_legendre(p)[0]
    pts, comps = [], []
    for c in range(3):
        n = [p + 1] * 3
        n[c] = p
        nodes = [cp, cp, cp]
        nodes[c] = op
        for k in range(n[2]):
            for j in range(n[1]):
                for i in range(n[0]):
                    pts.append([nodes[0][i], nodes[1][j], nodes[2][k]])
                    comps.append(c)
    pts, comps = np.array(pts), np.array(comps)
    J = mesh.jacobian_at(pts)  # [e, l, i, d]
    Bx, _ = _q2_1d(pts[:, 0])
    By, _ = _q2_1d(pts[:, 1])
    Bz, _ = _q2_1d(pts[:, 2])
    X = mesh.elem_coords().reshape(mesh.ne, 3, 3, 3, 3)
    xp = np.einsum("lk,lj,li,ekjic->elc", Bz, By, Bx, X)
    t = np.take_along_axis(J, comps[None, :, None, None].repeat(mesh.ne, 0).repeat(3, 2), axis=3)[..., 0]
    val = np.einsum("elc,elc->el", F(xp), t) * space.elem_sign_lex
    out = np.zeros(space.ndofs)
    out[space.elem_dof_lex] = val
    return out


def h1_interpolate(space, f):
    """Nodal interpolant of a scalar function f(x) in an H1HexSpace-like space (local vector)."""
    from palace_amd.fem.basis1d import gauss_lobatto
    from palace_amd.fem.mesh import _q2_1d

    p, mesh = space.p, space.mesh
    cp = gauss_lobatto(p + 1)
    pts = np.array([[cp[i], cp[j], cp[k]] for k in range(p + 1) for j in range(p + 1) for i in range(p + 1)])
    Bx, _ = _q2_1d(pts[:, 0])
    By, _ = _q2_1d(pts[:, 1])
    Bz, _ = _q2_1d(pts[:, 2])
    X = mesh.elem_coords().reshape(mesh.ne, 3, 3, 3, 3)
    xp = np.einsum("lk,lj,li,ekjic->elc", Bz, By, Bx, X)
    out = np.zeros(space.ndofs)
    out[space.elem_dof_lex] = f(xp)
    return out
