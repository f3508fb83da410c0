"""Generate tests/golden/qf_golden.npz by running the REFERENCE's own QFunction headers
(compiled into oracle/_ref by `make -C oracle ref`) on seeded random inputs.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
The fixtures pin oracle/palace_oracle.py and oracle/oracle_c.c on machines without the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from oracle import palace_oracle as po  # noqa: E402


def main():
    capi.build(ref=True)
    rng = np.random.default_rng(20260925)
    Q = 64
    # well-conditioned Jacobians: identity + perturbation, column-major [9][Q]
    J = (np.eye(3).reshape(9, 1) + 0.3 * rng.uniform(-1, 1, (9, Q))) * rng.uniform(0.5, 2.0, (1, Q))
    attr = rng.integers(1, 4, Q).astype(np.float64)  # 3 attributes
    qw = rng.uniform(0.01, 0.2, Q)
    geom = np.zeros((11, Q))
    capi.ref_call("f_build_geom_factor_33", None, Q, [attr, qw, np.ascontiguousarray(J)], [geom])
    # coefficient contexts: attr 1 -> anisotropic SPD, attr 2 -> scalar, attr 3 -> unassigned (zero)
    A = rng.uniform(-1, 1, (3, 3))
    aniso = A @ A.T + 3 * np.eye(3)
    ns = rng.uniform(-1, 1, (3, 3)) + 2 * np.eye(3)  # non-symmetric to expose transposition errors
    ctx_a = po.CoeffCtx(attr_mat=[0, 1, -1], mat_coeff=[aniso, np.array([2.08])])
    ctx_b = po.CoeffCtx(attr_mat=[1, 0, 1], mat_coeff=[ns, np.array([0.37])], a=1.5)
    ctx_id = po.CoeffCtx()
    u = rng.uniform(-1, 1, (3, Q))
    cu = rng.uniform(-1, 1, (3, Q))
    out = dict(Q=Q, J=J, attr=attr, qw=qw, geom=geom, u=u, cu=cu,
               ctx_a=ctx_a.pack(), ctx_b=ctx_b.pack(), ctx_id=ctx_id.pack(),
               ctx_pair=po.pack_pair(ctx_a, ctx_b))
    for tag, blob in (("a", ctx_a.pack()), ("b", ctx_b.pack()), ("id", ctx_id.pack())):
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hcurl_33", blob, Q, [geom, u], [v])
        out["hcurl_" + tag] = v
        v = np.zeros((3, Q))
        capi.ref_call("f_apply_hdiv_33", blob, Q, [geom, cu], [v])
        out["hdiv_" + tag] = v
    v, cv = np.zeros((3, Q)), np.zeros((3, Q))
    capi.ref_call("f_apply_hdivmass_33", po.pack_pair(ctx_a, ctx_b), Q, [geom, u, cu], [v, cv])
    out["hdivmass_v"], out["hdivmass_cv"] = v, cv
    np.savez(os.path.join(ROOT, "tests", "golden", "qf_golden.npz"), **out)
    print("wrote qf_golden.npz")


if __name__ == "__main__":
    main()


def mesh_fixture():
    """The reference's own cylinder input (examples/cylinder/mesh/cylinder_hex.msh, 80 hex27)
    converted to arrays so GPU-box tests (no /root/reference there) run on the real mesh."""
    from palace_amd.fem.mesh import read_gmsh22

    m = read_gmsh22("/root/reference/examples/cylinder/mesh/cylinder_hex.msh")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cylinder_hex_mesh.npz"),
                        x=m.x, elem_nodes=m.elem_nodes.astype(np.int32), attr=m.attr,
                        bdr_faces=m.bdr_faces.astype(np.int32), bdr_attr=m.bdr_attr)
    print("wrote cylinder_hex_mesh.npz")


if __name__ == "__main__":
    mesh_fixture()
