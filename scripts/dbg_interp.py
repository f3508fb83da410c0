import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from palace_amd import linalg
from palace_amd.fem import tet
from oracle import palace_oracle as po
def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)
ctx = linalg.Context()
mesh = tet.to_quadratic(tet.cube_tet_mesh(3), _warp)
for pc, pf in ((1, 2), (2, 3)):
    c, f = tet.NDTetSpace(mesh, pc), tet.NDTetSpace(mesh, pf)
    M = tet.nd_tet_transfer_matrix(pc, pf)
    P = linalg.DenseInterp(ctx, c.restriction(), f.restriction(interp_range=True), M)
    os.environ["PALACE_AMD_DENSE_INTERP"] = "lds"
    P0 = linalg.DenseInterp(ctx, c.restriction(), f.restriction(interp_range=True), M)
    del os.environ["PALACE_AMD_DENSE_INTERP"]
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.uniform(-1, 1, c.ndofs)).cuda()
    z = torch.from_numpy(rng.uniform(-1, 1, f.ndofs)).cuda()
    y, y0 = torch.empty_like(z), torch.empty_like(z)
    P.mult(x, y); P0.mult(x, y0)
    t, t0 = torch.empty_like(x), torch.empty_like(x)
    P.mult_transpose(z, t); P0.mult_transpose(z, t0)
    print(pc, pf, "fwd diff", float((y - y0).abs().max()), "tr diff", float((t - t0).abs().max()), "ne", mesh.ne, "nbad", int(((t - t0).abs() > 1e-12).sum()), "of", c.ndofs)
    bad = torch.nonzero((t - t0).abs() > 1e-12).flatten()[:10].cpu().numpy()
    print(" bad dofs", bad, (t - t0)[bad].cpu().numpy(), t0[bad].cpu().numpy())
