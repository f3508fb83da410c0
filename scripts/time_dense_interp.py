"""Event timings of the dense element interpolators of a tetrahedral p-multigrid (prolongation ND p-1 -> p and its transpose, the
discrete gradient H1 p -> ND p) on `n`^3 x 6 tetrahedra (default 36: 279 936).  A/B of pa_interp.hip's dense_interp_kernel."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from palace_amd import linalg  # noqa: E402
from palace_amd.fem import tet  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    mesh = tet.cube_tet_mesh(n)
    ctx = linalg.Context()
    nd = {p: tet.NDTetSpace(mesh, p) for p in (1, 2, 3)}
    h1 = tet.H1TetSpace(mesh, 3)
    ops = {"P 1->2": (linalg.DenseInterp(ctx, nd[1].restriction(), nd[2].restriction(interp_range=True), tet.nd_tet_transfer_matrix(1, 2)), nd[1].ndofs, nd[2].ndofs),
           "P 2->3": (linalg.DenseInterp(ctx, nd[2].restriction(), nd[3].restriction(interp_range=True), tet.nd_tet_transfer_matrix(2, 3)), nd[2].ndofs, nd[3].ndofs),
           "G p=3": (linalg.DenseInterp(ctx, h1.restriction(), nd[3].restriction(interp_range=True), tet.tet_gradient_matrix(3)), h1.ndofs, nd[3].ndofs)}
    print("%d tetrahedra" % mesh.ne)
    for name, (op, nd_, nr) in ops.items():
        x = torch.rand(nd_, dtype=torch.float64, device="cuda")
        y = torch.empty(nr, dtype=torch.float64, device="cuda")
        z = torch.empty(nd_, dtype=torch.float64, device="cuda")
        for f, a, b in ((op.mult, x, y), (op.mult_transpose, y, z)):
            for _ in range(5):
                f(a, b)
            ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            import time
            best = 1e9  # (best of five blocks of 20: one host hiccup does not become the number)
            for _blk in range(5):
                t0 = time.perf_counter()
                for _ in range(20):
                    f(a, b)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20)
            dt = best
            print("%-8s %-10s %8.1f us  (%d -> %d dofs)  checksum %.12e" % (name, f.__name__, dt * 1e6, a.numel(), b.numel(), float(b.double().sum())))


if __name__ == "__main__":
    main()
