"""Write the descriptor arrays of a small PEC-cavity problem (what Palace's libCEED glue would pass across
the C ABI: restriction, 1-D tables, mesh nodes, coefficient context, essential dofs) into one binary file
for the C++ host example.  Usage: python dump_problem.py out.bin [order]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palace_amd import ceed  # noqa: E402
from palace_amd.fem.basis1d import Tables1D, gauss_legendre  # noqa: E402
from palace_amd.fem.fespace import NDHexSpace  # noqa: E402
from palace_amd.fem.mesh import _q2_1d, ogrid_cylinder  # noqa: E402


def main(path, p=2):
    mesh = ogrid_cylinder(2, 4)
    nd = NDHexSpace(mesh, p)
    q1d = p + 1
    t = Tables1D(p, q1d)
    off, ori = nd.native_restriction()
    qx, qw = gauss_legendre(q1d)
    B, G = _q2_1d(qx)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    curl = ceed.coefficient_context(3)
    arrays = [np.array([mesh.ne, nd.P, nd.ndofs, p, q1d, mesh.x.shape[0]], dtype=np.int32),
              off.astype(np.int32), ori.astype(np.uint8), np.asarray(nd.dof_map_native(), dtype=np.int32),
              t.Bc, t.Gc, t.Bo, mesh.elem_nodes.astype(np.int32), mesh.x.astype(np.float64), mesh.attr.astype(np.int32),
              B, G, qw, np.concatenate([mass, curl]), nd.ess_dofs().astype(np.int32)]
    with open(path, "wb") as f:
        f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
        for a in arrays:
            a = np.ascontiguousarray(a)
            f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
            f.write(a.tobytes())


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
