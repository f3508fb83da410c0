"""CPU: time the host set-up of the native AMS / AMG hierarchy (palace_amd/csrc/amg.hip + the Galerkin products of
amg_solver.hip) on the level-0 matrix of the bench problem -- the order-1 Nedelec curl-curl + mass matrix of the 125 440-element
cylinder (0.38M rows), its discrete gradient and vertex coordinates, assembled here through the oracle's element matrices --
without a GPU.  Usage: python scripts/time_amg_setup.py [target_dofs_at_p3=1e7] [threads]
(writes the matrices to /tmp/amg/level0.bin once, builds scripts/time_amg_setup.cpp with hipcc, host code only)."""
import os
import subprocess
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_inputs(target, path):
    from oracle import palace_oracle as po
    from palace_amd.fem.fespace import H1HexSpace, NDHexSpace, lowest_order_gradient, vertex_coordinates
    from palace_amd.fem.mesh import cylinder_for_dofs
    from tests import util

    mesh = cylinder_for_dofs(target, 3)
    nd, h1 = NDHexSpace(mesh, 1), H1HexSpace(mesh, 1)
    off, ori = nd.native_restriction()
    interp, curl = po.nd_hex_dense_tables(1, 4, nd.dof_map_native())  # the fine rule (Q1 = 4), as CeedOperatorCoarsen keeps it
    mass, cc = po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([2.08])]), po.CoeffCtx()
    rows, cols, vals = [], [], []
    for a in range(0, mesh.ne, 8192):
        sl = slice(a, min(mesh.ne, a + 8192))
        geom = util.oracle_geom(_Sub(mesh, sl), 4)
        orc = po.CeedOperatorOracle(nd.ndofs, off[sl], ori[sl], interp, curl, geom, po.QF_HDIVMASS, mass, cc)
        Ae = orc.element_matrices()
        o = off[sl].astype(np.int64)
        rows.append(np.repeat(o, o.shape[1], axis=1).ravel())
        cols.append(np.tile(o, (1, o.shape[1])).ravel())
        vals.append(Ae.ravel())
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd.ndofs, nd.ndofs)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    ess = np.zeros(nd.ndofs, dtype=np.int8)
    ess[nd.ess_dofs()] = 1
    # essential rows / columns eliminated, unit diagonal (ParOperator DIAG_ONE on the assembled level)
    keep = sp.diags(1.0 - ess.astype(np.float64))
    A = (keep @ A @ keep + sp.diags(ess.astype(np.float64))).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    G = lowest_order_gradient(h1, nd).tocsr()
    G.sort_indices()
    xyz = np.ascontiguousarray(vertex_coordinates(h1), dtype=np.float64)
    with open(path, "wb") as f:
        for M in (A, G):
            np.array([M.shape[0], M.shape[1], M.nnz], dtype=np.int64).tofile(f)
            M.indptr.astype(np.int32).tofile(f)
            M.indices.astype(np.int32).tofile(f)
            M.data.astype(np.float64).tofile(f)
        np.array([xyz.shape[0]], dtype=np.int64).tofile(f)
        xyz.tofile(f)
        ess.tofile(f)
    print("level 0: %d rows, %d nnz; G %d x %d" % (A.shape[0], A.nnz, G.shape[0], G.shape[1]))


class _Sub:
    """A slice of a HexMesh for tests.util.oracle_geom (elem_coords, attr, ne)."""

    def __init__(self, mesh, sl):
        self._c, self.attr = mesh.elem_coords()[sl], mesh.attr[sl]
        self.ne = self._c.shape[0]

    def elem_coords(self):
        return self._c


def main():
    target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
    threads = sys.argv[2] if len(sys.argv) > 2 else ""
    path = "/tmp/amg/level0_%g.bin" % target
    os.makedirs("/tmp/amg", exist_ok=True)
    if not os.path.exists(path):
        build_inputs(target, path)
    exe = "/tmp/amg/time_amg_setup"
    csrc = os.path.join(ROOT, "palace_amd", "csrc")
    lib = os.path.join(ROOT, "palace_amd", "lib")  # (the harness calls the set-up functions of the in-tree library)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-w", "-I" + csrc, os.path.join(ROOT, "scripts", "time_amg_setup.cpp"),
                           "-L" + lib, "-lpalace_amd", "-Wl,-rpath," + lib, "-o", exe])
    env = dict(os.environ)
    if threads:
        env["PALACE_AMD_SETUP_THREADS"] = threads
    subprocess.check_call([exe, path], env=env)


if __name__ == "__main__":
    main()
