"""h-refinement transfers: the prolongation between the spaces of one finite element collection on a mesh and on its uniform
refinement -- what the reference wraps as mfem::TransferOperator for two multigrid levels on DIFFERENT meshes
(fem/fespace.cpp:246-251; the h-levels of ConstructFiniteElementSpaceHierarchy, fem/multigrid.hpp:103-112, utils/geodata.cpp:
426-460 keeps every uniformly refined mesh as a level).  Host side only: the element matrices (MFEM's
FiniteElement::GetLocalInterpolation of a child's embedding into its parent) and the parent's dofs per fine element; the
apply is pa_interp_create_refinement (pa_interp.hip: dense_interp_kernel with one matrix per child type)."""
from __future__ import annotations

import numpy as np

from .basis1d import gauss_legendre, gauss_lobatto, lagrange_eval


def hex_child_matrices(p: int, hcurl: bool = True) -> np.ndarray:
    """[8][P, P]: local interpolation of child a + 2 b + 4 c (its octant of the parent's reference cube: the child order of
    mesh.refine_uniform) for the order-p tensor element, tensor (lexicographic) dof order.  Covariant elements: the tangent of a
    fine dof is half as long in the parent's coordinates (factor 1/2 in the open direction)."""
    cp = gauss_lobatto(p + 1)
    op = gauss_legendre(p)[0] if hcurl else None
    one_d = lambda nodes, half, scale: scale * lagrange_eval(nodes, 0.5 * (half + nodes))[0]  # noqa: E731  [child node, parent fn]
    out = []
    for c in range(2):
        for b in range(2):
            for a in range(2):
                h = (a, b, c)
                if not hcurl:
                    out.append(np.kron(one_d(cp, h[2], 1.0), np.kron(one_d(cp, h[1], 1.0), one_d(cp, h[0], 1.0))))
                    continue
                blocks = []
                for comp in range(3):
                    m1 = [one_d(op, h[d], 0.5) if d == comp else one_d(cp, h[d], 1.0) for d in range(3)]
                    blocks.append(np.kron(m1[2], np.kron(m1[1], m1[0])))
                n = blocks[0].shape[0]
                M = np.zeros((3 * n, 3 * n))
                for comp in range(3):
                    M[comp * n:(comp + 1) * n, comp * n:(comp + 1) * n] = blocks[comp]
                out.append(M)
    return np.ascontiguousarray(out)


def hex_refinement(coarse_space, fine_space):
    """(dom, rng, M, mat_id) for linalg.RefinementTransfer between an NDHexSpace / H1HexSpace on a HexMesh and the space of
    the same order on mesh.refine_uniform(mesh): fine element 8 E + k is child k of E."""
    assert coarse_space.p == fine_space.p and type(coarse_space) is type(fine_space)
    ne_f = fine_space.mesh.ne
    assert ne_f == 8 * coarse_space.mesh.ne, "the fine mesh is not one uniform refinement of the coarse mesh"
    parent = np.arange(ne_f) // 8
    hcurl = hasattr(coarse_space, "elem_sign_lex")
    dom = dict(offsets=np.ascontiguousarray(coarse_space.elem_dof_lex[parent]), lsize=coarse_space.ndofs)
    rng = dict(offsets=np.ascontiguousarray(fine_space.elem_dof_lex), lsize=fine_space.ndofs)
    if hcurl:
        dom["orients"] = np.ascontiguousarray(coarse_space.elem_sign_lex[parent] < 0)
        rng["orients"] = np.ascontiguousarray(fine_space.elem_sign_lex < 0)
    return dom, rng, hex_child_matrices(coarse_space.p, hcurl), (np.arange(ne_f) % 8).astype(np.uint8)


def tet_refinement(coarse_space, fine_space):
    """The same for NDTetSpace / H1TetSpace of order 1 on a TetMesh and on tet.refine_uniform(mesh) (children 8 E + k; their
    vertex order -- and so their embedding -- depends on the orientation fix of the refinement: the embeddings are recovered
    from the vertex coordinates and grouped)."""
    from . import tet

    mc, mf = coarse_space.mesh, fine_space.mesh
    assert coarse_space.p == 1 and fine_space.p == 1 and mf.ne == 8 * mc.ne
    parent = np.arange(mf.ne) // 8
    Xp = mc.verts[mc.tets[parent]]                                   # [ne_f, 4, 3]
    Xc = mf.verts[mf.tets]
    Ap = np.transpose(Xp[:, 1:] - Xp[:, :1], (0, 2, 1))              # columns: parent edge vectors
    ref = np.linalg.solve(Ap[:, None], (Xc - Xp[:, :1])[..., None])[..., 0]   # child vertices in the parent's reference coordinates
    ref = np.round(2.0 * ref) / 2.0
    assert np.abs(np.einsum("eij,evj->evi", Ap, ref) + Xp[:, :1] - Xc).max() < 1e-9 * np.abs(Xp).max()
    keys, mat_id = np.unique(ref.reshape(mf.ne, 12), axis=0, return_inverse=True)
    hcurl = isinstance(coarse_space, tet.NDTetSpace)
    Ms = []
    for k in keys:
        v = k.reshape(4, 3)
        o, A = v[0], (v[1:] - v[0]).T                                # x_parent = o + A x_child
        if hcurl:
            el = coarse_space.elem
            val, _ = el.tables(o[None, :] + el.dof_pts @ A.T)        # [3, P_f, P_c] parent basis at the child's dof points
            Ms.append(np.einsum("dkj,kd->kj", val, el.dof_tans @ A.T))
        else:
            el = coarse_space.elem
            val, _ = el.tables(o[None, :] + tet.h1_tet_nodes(1) @ A.T)
            Ms.append(val[0])
    dom = dict(offsets=np.ascontiguousarray(coarse_space.offsets[parent]), lsize=coarse_space.ndofs)
    rng = dict(offsets=np.ascontiguousarray(fine_space.offsets), lsize=fine_space.ndofs)
    if hcurl:
        dom["orients"] = np.ascontiguousarray(coarse_space.orients[parent])
        rng["orients"] = np.ascontiguousarray(fine_space.orients)
    return dom, rng, np.ascontiguousarray(Ms), mat_id.astype(np.uint8)
