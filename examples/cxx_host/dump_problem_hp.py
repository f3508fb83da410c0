"""The arrays Palace's MFEM side would hand over for a PEC cavity with the FULL multigrid hierarchy of the reference -- h-levels
on a sequence of uniformly refined meshes under the p-levels of the finest mesh (fem/multigrid.hpp:77-123, utils/geodata.cpp:
426-460) -- in one binary file for solve_hp.cpp: every mesh of the sequence (nodes, element node lattices, attributes) with, from
the second on, its mfem::Mesh::GetRefinementTransforms() data (parent element, point matrix index, the eight point matrices),
and per level the element -> dof tables of the Nedelec and H1 spaces, tensor -> native maps and essential (PEC) dofs.
Usage: python dump_problem_hp.py out.bin [order] [h_levels] [n] [nz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palace_amd.fem.fespace import H1HexSpace, NDHexSpace  # noqa: E402
from palace_amd.fem.mesh import ogrid_cylinder, refine_uniform  # noqa: E402
from palace_amd.fem.partition import levels_for  # noqa: E402


def main(path, p=2, h_levels=1, n=2, nz=3):
    meshes = [ogrid_cylinder(n, nz)]
    for _ in range(h_levels):
        meshes.append(refine_uniform(meshes[-1]))
    orders = levels_for(p)
    levels = [(m, orders[0]) for m in range(h_levels)] + [(h_levels, q) for q in orders]
    arrays = [np.array([len(meshes), p, len(levels)] + [v for lv in levels for v in lv], dtype=np.int32)]
    # child a + 2 b + 4 c of a hexahedron: its corners (lexicographic) in the parent's reference cube
    pm = np.zeros((8, 8, 3))
    for k in range(8):
        o = 0.5 * np.array([k & 1, (k >> 1) & 1, k >> 2])
        for v in range(8):
            pm[k, v] = o + 0.5 * np.array([v & 1, (v >> 1) & 1, v >> 2])
    for i, mesh in enumerate(meshes):
        arrays += [np.array([mesh.ne, mesh.x.shape[0]], dtype=np.int32), mesh.elem_nodes.astype(np.int32), mesh.x.astype(np.float64),
                   mesh.attr.astype(np.int32)]
        if i > 0:
            arrays += [(np.arange(mesh.ne) // 8).astype(np.int32), (np.arange(mesh.ne) % 8).astype(np.int32), pm]
    for m, q in levels:
        nd, h1 = NDHexSpace(meshes[m], q), H1HexSpace(meshes[m], q)
        off, ori = nd.native_restriction()
        arrays += [np.array([nd.ndofs, h1.ndofs], dtype=np.int32), off.astype(np.int32), ori.astype(np.uint8),
                   np.asarray(nd.dof_map_native(), dtype=np.int32), nd.ess_dofs().astype(np.int32),
                   h1.elem_dof_lex.astype(np.int32), h1.ess_dofs().astype(np.int32)]
    with open(path, "wb") as f:
        f.write(np.array([len(arrays)], dtype=np.int64).tobytes())
        for a in arrays:
            a = np.ascontiguousarray(a)
            f.write(np.array([a.nbytes], dtype=np.int64).tobytes())
            f.write(a.tobytes())


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[2:]]
    main(sys.argv[1], *a)
