"""Arrays for the plane C++ estimator example (estimate2d.cpp): the triangles of the reference's cavity2d mesh, its Nedelec
space, the rotated-Nedelec form of the Raviart-Thomas space, the H1 space and a discontinuous scalar space of the same order
(the space the scalar curl of a plane field lives in), a 2 x 2 permittivity and a scalar curl-curl inverse permeability per
attribute, and the fields E (ND) and B_z (scalar) to estimate.
Usage: python dump_estimator_problem_2d.py out.bin [p]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palace_amd.fem import tri  # noqa: E402


def problem(p=2):
    M_ = np.load(os.path.join(ROOT, "tests", "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    attr = 1 + (np.arange(en.shape[0]) % 2)
    m = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), attr, elem_nodes=en, nodes=M_["nodes"])
    nd, h1 = tri.NDTriSpace(m, p), tri.H1TriSpace(m, p)
    pts, wts = tri.tri_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    hint, hgrad = h1.elem.tables(pts)
    rint = np.stack([nint[1], -nint[0]])  # u_RT = R u_ND: contravariant image = rotated covariant image (J R = det J R J^-T)
    l2_off = np.arange(m.ne * h1.P, dtype=np.int32).reshape(m.ne, h1.P)
    eps = [np.array([[2.0, 0.3], [0.3, 1.5]]), np.eye(2) * 3.1]
    muinv = [np.array([[0.8]]), np.array([[1.4]])]
    rng = np.random.default_rng(23)
    E, B = rng.uniform(-1, 1, nd.ndofs), rng.uniform(-1, 1, l2_off.size)
    return dict(mesh=m, nd=nd, h1=h1, pts=pts, wts=wts, nint=nint, ncurl=ncurl, rint=rint, hint=hint, hgrad=hgrad, l2_off=l2_off,
                eps=eps, muinv=muinv, E=E, B=B)


def main(path, p=2):
    P = problem(p)
    m, nd, h1 = P["mesh"], P["nd"], P["h1"]
    G = m.geometry_grad_table(P["pts"])
    arrays = [np.array([m.ne, m.elem_nodes.shape[1], len(P["wts"]), m.nodes.shape[0], p, nd.ndofs, nd.P, h1.ndofs, h1.P,
                        P["l2_off"].size], dtype=np.int32),
              m.elem_nodes.astype(np.int32), m.nodes.astype(np.float64), m.attr.astype(np.int32), np.asarray(G, np.float64),
              np.asarray(P["wts"], np.float64),
              nd.offsets.astype(np.int32), np.asarray(nd.orients, np.uint8), np.asarray(P["nint"], np.float64),
              np.asarray(P["ncurl"], np.float64), np.asarray(P["rint"], np.float64),
              h1.offsets.astype(np.int32), np.asarray(P["hint"], np.float64), np.asarray(P["hgrad"], np.float64),
              P["l2_off"], np.concatenate([e.T.ravel() for e in P["eps"]]), np.concatenate([e.ravel() for e in P["muinv"]]),
              P["E"], P["B"]]
    with open(path, "wb") as f:
        f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
        for a in arrays:
            a = np.ascontiguousarray(a)
            f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
            f.write(a.tobytes())


if __name__ == "__main__":
    main(sys.argv[1], *[int(a) for a in sys.argv[2:]])
