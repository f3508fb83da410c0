import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from palace_amd import ceed
from palace_amd.fem.fespace import NDHexSpace
from palace_amd.fem.mesh import HexMesh
from tests import util
d = np.load("tests/golden/cylinder_hex_mesh.npz")
mesh = HexMesh(x=d["x"], elem_nodes=d["elem_nodes"].astype(np.int64), attr=d["attr"])
for (pc, pf) in [(2,5-1),(3,4),(2,4)]:
    q1d = pf+1
    ndf, ndc = NDHexSpace(mesh, pf), NDHexSpace(mesh, pc)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_s = util.make_ctx("scalar"); _, b_i = util.make_ctx("identity")
    for name, mk in (("mass", lambda: ceed.ndmass_operator(geom, ndc, b_s)), ("curl", lambda: ceed.curlcurl_operator(geom, ndc, b_i))):
        op = mk()
        x = np.random.default_rng(2).uniform(-1, 1, ndc.ndofs)
        ys = []
        for rep in range(2):
            y = op.mult(torch.from_numpy(x).cuda(), torch.empty(ndc.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy(); ys.append(y)
        ref = util.oracle_apply_c(ndc, util.oracle_geom(mesh, q1d), "hcurl" if name=="mass" else "hdiv", b_s if name=="mass" else b_i, x, q1d)
        err = np.abs(ys[0]-ref)
        bad = np.nonzero(err > 1e-10*np.abs(ref).max())[0]
        print(pc, pf, name, "rel", np.linalg.norm(ys[0]-ref)/np.linalg.norm(ref), "determ", np.abs(ys[0]-ys[1]).max(), "nbad", bad.size, "of", ndc.ndofs,
              "edge/face/int bases", ndc.face_base, ndc.int_base, "bad range", (bad.min(), bad.max()) if bad.size else None)
        if bad.size:
            # which lex local dofs are bad? apply on single element
            e_bad = {}
            for b in bad[:2000]:
                es, ls = np.nonzero(ndc.elem_dof_lex == b)
                for l in ls: e_bad[l] = e_bad.get(l,0)+1
            print(sorted(e_bad.items()))
