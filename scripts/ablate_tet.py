"""Stage ablation of the resident dense kernel in ONE process (the ablation library, `make -C palace_amd/csrc ablate`):
PA_DBG bits: 1 no x gather, 2 q-data from one block, 4 no MFMA work, 8 E-vector stores to one slot, 16 no curl-orientation exchange.
  PALACE_AMD_LIB=$PWD/palace_amd/lib/libpalace_amd_ablate.so python scripts/ablate_tet.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from palace_amd import ceed
from palace_amd.fem import tet
n = int(os.environ.get("N", "36")); p = int(os.environ.get("P", "3")); reps = int(os.environ.get("REPS", "30"))
mesh = tet.cube_tet_mesh(n)
nd = tet.NDTetSpace(mesh, p)
pts, wts = tet.default_tet_rule(p)
interp, curl = nd.elem.tables(pts)
geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
ident = ceed.coefficient_context(3)
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
ops = {"curl": ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIV_33, ident, ceed.EVAL_CURL).finalize(),
       "curlmass": ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([mass, ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()}
print(f"{mesh.ne} tets, p={p}, {nd.ndofs} dofs; affine: {{k: o.dense_affine() for k, o in ops.items()}}", flush=True)
x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for dbg in (0, 1, 2, 4, 8, 16, 5, 12, 13, 29, 31, 0):
    os.environ["PA_DBG"] = str(dbg)
    row = []
    for name, op in ops.items():
        for _ in range(5): op.mult(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): op.mult(x, y)
        e1.record(); torch.cuda.synchronize()
        row.append(f"{name} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us")
    print(f"PA_DBG={dbg:2d}  " + "   ".join(row), flush=True)
