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
    else:
        raise ValueError(kind)
    return c, c.pack()


QF_MAP = {"hdiv": (po.QF_HDIV, capi.QF_HDIV), "hcurl": (po.QF_HCURL, capi.QF_HCURL),
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
