"""Arrays for the C++ estimator example (estimate.cpp): a tetrahedral mesh (straight or curved), its Nedelec and
Raviart-Thomas spaces as Palace hands them to libCEED on the non-tensor path (native restrictions with sign flips or the
tridiagonal curl-orientation, dense value / curl tables at the quadrature points, fem/libceed/{restriction,basis}.cpp), two
anisotropic material tensors and the fields E (ND) and B (RT) to estimate.
Usage: python dump_estimator_problem.py out.bin [p] [n] [curved=0|1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palace_amd.fem import rt, tet  # noqa: E402


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def problem(p=2, n=3, curved=0):
    m = tet.cube_tet_mesh(n)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    if curved:
        m2 = tet.to_quadratic(m, _warp)
        m2.attr[:] = m.attr
        m = m2
    nd, sp, h1 = tet.NDTetSpace(m, p), rt.RTTetSpace(m, p), tet.H1TetSpace(m, p)
    pts, wts = tet.tet_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    rint, rdiv = sp.elem.tables(pts)
    hint, hgrad = h1.elem.tables(pts)
    eps = [np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.1], [0.0, 0.1, 1.2]]), np.diag([1.0, 1.0, 1.0]) * 3.1]
    muinv = [np.diag([0.8, 1.1, 0.9]), np.array([[1.4, -0.2, 0.1], [-0.2, 1.0, 0.0], [0.1, 0.0, 0.7]])]
    rng = np.random.default_rng(17)
    E, B = rng.uniform(-1, 1, nd.ndofs), rng.uniform(-1, 1, sp.ndofs)
    phi = rng.uniform(-1, 1, h1.ndofs)
    return dict(mesh=m, nd=nd, rt=sp, pts=pts, wts=wts, nint=nint, ncurl=ncurl, rint=rint, eps=eps, muinv=muinv, E=E, B=B,
                h1=h1, hint=hint, hgrad=hgrad, phi=phi, rdiv=rdiv)


def main(path, p=2, n=3, curved=0):
    P = problem(p, n, curved)
    m, nd, sp = P["mesh"], P["nd"], P["rt"]
    G = m.geometry_grad_table(P["pts"])
    cor = nd.curl_orients if not nd.diagonal_transform else np.zeros(0, np.int8)
    ori = nd.orients if nd.diagonal_transform else np.zeros(0, np.uint8)
    arrays = [np.array([m.ne, m.elem_nodes.shape[1], len(P["wts"]), m.nodes.shape[0], p, nd.ndofs, nd.P, sp.ndofs, sp.P,
                        int(nd.diagonal_transform)], dtype=np.int32),
              m.elem_nodes.astype(np.int32), m.nodes.astype(np.float64), m.attr.astype(np.int32), np.asarray(G, np.float64),
              np.asarray(P["wts"], np.float64),
              nd.offsets.astype(np.int32), np.asarray(ori, np.uint8), np.asarray(cor, np.int8), np.asarray(P["nint"], np.float64),
              np.asarray(P["ncurl"], np.float64),
              sp.offsets.astype(np.int32), np.asarray(sp.orients, np.uint8), np.asarray(P["rint"], np.float64),
              np.concatenate([e.T.ravel() for e in P["eps"]]), np.concatenate([e.T.ravel() for e in P["muinv"]]),
              P["E"], P["B"],
              # the H1 space of the same order and a potential: MixedVectorGradientIntegrator (eps grad phi, v) into ND and RT
              np.array([P["h1"].ndofs, P["h1"].P], dtype=np.int32), P["h1"].offsets.astype(np.int32),
              np.asarray(P["hint"], np.float64), np.asarray(P["hgrad"], np.float64), P["phi"],
              # the divergence table of the RT space: DivDivMassIntegrator (muinv B, v) + (c div B, div v) through BilinearForm(rt)
              np.asarray(P["rdiv"], np.float64)]
    with open(path, "wb") as f:
        f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
        for a in arrays:
            a = np.ascontiguousarray(a)
            f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
            f.write(a.tobytes())


if __name__ == "__main__":
    main(sys.argv[1], *[int(v) for v in sys.argv[2:]])
