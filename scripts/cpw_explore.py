"""FGMRES iteration counts of the cpw driven system (bench.py's cpw leg) against the preconditioner settings.
  python scripts/cpw_explore.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from palace_amd import linalg
from palace_amd.fem import tet
from palace_amd.fem.tetproblem import TetProblem

d = np.load("tests/golden/cpw_mesh.npz")
base = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
k0 = 2 * np.pi * 16.0e9 * 1.0e-6 / 299792458.0
for refine in (0, 1):
    mesh = base
    for _ in range(refine):
        mesh = tet.refine_uniform(mesh)
    bt = np.sort(np.asarray(mesh.bdr_tris, dtype=np.int64), axis=1)
    pec = bt[np.isin(mesh.bdr_attr, (4, 13))]
    fv, nvt = mesh.face_verts, mesh.nv
    key = lambda f: (f[:, 0] * nvt + f[:, 1]) * nvt + f[:, 2]
    of = np.argsort(key(fv))
    fmask = np.zeros(fv.shape[0], dtype=bool)
    fmask[of[np.searchsorted(key(fv)[of], key(pec))]] = True
    for cheby, coarse, smooth_rhs in ((4, "ams", False), (6, "ams", False), (6, "cg", False), (6, "ams", True)):
        prob = TetProblem(linalg.Context(), mesh, 3)
        s = prob.driven_solver(fmask, k0, eps=[1.0, 11.7], tand=[0.0, 0.05], coarse=coarse, cheby_order=cheby, max_it=600,
                               restart=600, coarse_tol=1e-3)
        n = s["n"]
        rng = np.random.default_rng(4)
        if smooth_rhs:  # b = A x for a smooth x instead of white noise
            xs = np.ones(n) + 0.5j * np.ones(n)
            xs[s["ess"]] = 0.0
            br, bi = s["A"].mult(torch.from_numpy(xs.real.copy()).cuda(), torch.from_numpy(xs.imag.copy()).cuda(),
                                 torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda"))
            br[torch.from_numpy(s["ess"].astype(np.int64)).cuda()] = 0.0
            bi[torch.from_numpy(s["ess"].astype(np.int64)).cuda()] = 0.0
        else:
            b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
            b[s["ess"]] = 0.0
            br, bi = torch.from_numpy(b.real.copy()).cuda(), torch.from_numpy(b.imag.copy()).cuda()
        xr, xi = torch.zeros_like(br), torch.zeros_like(br)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s["solver"].mult(br, bi, xr, xi)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = s["solver"].stats()
        print(f"refine {refine} n {n} cheby {cheby} coarse {coarse} smooth_rhs {smooth_rhs}: its {st['iterations']} conv {st['converged']} "
              f"rel {st['final_res'] / st['initial_res']:.2e}  {dt:.2f} s  {st['iterations'] / dt:.1f} it/s", flush=True)
        del prob, s
