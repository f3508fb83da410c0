"""One file per rank for examples/cxx_host/solve_ranks.cpp: the z-slab of a PEC cylinder cavity a rank owns (mesh nodes, element ->
dof tables of the Nedelec and H1 spaces of every multigrid level in the rank's local numbering -- true dofs first, ghosts last --,
essential true dofs) and the halo plan of every space (neighbour ranks, owned dofs to send, ghost slots to receive): what Palace's
ParMesh / ParFiniteElementSpace / GroupCommunicator hold on a rank.  Usage: python dump_problem_ranks.py prefix world [order] [n] [nz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palace_amd.fem.partition import SlabH1Space, SlabProblem  # noqa: E402


def plan_arrays(space):
    nbr = np.asarray(space.nbr, dtype=np.int32)
    so = np.zeros(len(nbr) + 1, dtype=np.int32)
    ro = np.zeros(len(nbr) + 1, dtype=np.int32)
    so[1:] = np.cumsum([len(q) for q in space.send])
    ro[1:] = np.cumsum([len(q) for q in space.recv])
    si = np.concatenate(space.send).astype(np.int32) if len(nbr) else np.zeros(0, np.int32)
    ri = np.concatenate(space.recv).astype(np.int32) if len(nbr) else np.zeros(0, np.int32)
    return [nbr, so, si, ro, ri]


def main(prefix, world, p=2, n=2, nz=4):
    assert nz % world == 0
    for rank in range(world):
        prob = SlabProblem(None, rank, world, p, 0, shape=(n, nz // world), device=False)
        mesh, orders = prob.mesh, prob.orders
        z_lo = rank * prob.height
        arrays = [np.array([mesh.ne, mesh.x.shape[0], p, len(orders)] + orders, dtype=np.int32),
                  mesh.elem_nodes.astype(np.int32), mesh.x.astype(np.float64), mesh.attr.astype(np.int32)]
        for q, nd in zip(orders, prob.spaces):
            h1 = SlabH1Space(mesh, q, rank, world, z_lo, z_lo + prob.height, prob.radius)
            off, ori = nd.native_restriction()
            arrays += [np.array([nd.ndofs, h1.ndofs, nd.n_true, h1.n_true], dtype=np.int32), off.astype(np.int32), ori.astype(np.uint8),
                       np.asarray(nd.dof_map_native(), dtype=np.int32), nd.ess_dofs().astype(np.int32),
                       h1.elem_dof_lex.astype(np.int32), h1.ess_dofs().astype(np.int32)]
            arrays += plan_arrays(nd) + plan_arrays(h1)
        with open(f"{prefix}.{rank}", "wb") as f:
            f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
            for a in arrays:
                a = np.ascontiguousarray(a)
                f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
                f.write(a.tobytes())


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[3:]]
    main(sys.argv[1], int(sys.argv[2]), *a)
