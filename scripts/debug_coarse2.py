import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from palace_amd import ceed
from palace_amd.fem.fespace import NDHexSpace
from palace_amd.fem.mesh import HexMesh
from tests import util
d = np.load("tests/golden/cylinder_hex_mesh.npz")
mesh = HexMesh(x=d["x"], elem_nodes=d["elem_nodes"].astype(np.int64), attr=d["attr"])
for (pc, pf) in [(2,4),(3,4),(1,4),(4,4)]:
    q1d = pf+1
    ndc = NDHexSpace(mesh, pc)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_s = util.make_ctx("scalar"); _, b_i = util.make_ctx("identity")
    op = ceed.curlcurlmass_operator(geom, ndc, b_s, b_i)
    x = np.random.default_rng(2).uniform(-1, 1, ndc.ndofs)
    ys = []
    for rep in range(2):
        y = op.mult(torch.from_numpy(x).cuda(), torch.empty(ndc.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy(); ys.append(y)
    ref = util.oracle_apply_c(ndc, util.oracle_geom(mesh, q1d), "hdivmass", np.concatenate([b_s,b_i]), x, q1d)
    err = np.abs(ys[0]-ref)
    bad = np.nonzero(err > 1e-10*np.abs(ref).max())[0]
    print(os.environ.get("PALACE_AMD_LIB","default")[-20:], pc, pf, "rel", np.linalg.norm(ys[0]-ref)/np.linalg.norm(ref), "determ", np.abs(ys[0]-ys[1]).max(), "nbad", bad.size, "of", ndc.ndofs)
    if bad.size:
        e_bad = {}
        for b in bad[:3000]:
            es, ls = np.nonzero(ndc.elem_dof_lex == b)
            for l in ls: e_bad[int(l)] = e_bad.get(int(l),0)+1
        print(sorted(e_bad.items()))
        print("bad dof vals", ys[0][bad[:5]], ref[bad[:5]])
