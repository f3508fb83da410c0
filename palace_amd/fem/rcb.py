"""General element partition: recursive coordinate bisection of the element centroids, and the rank-local view of a
finite element space under any element partition (local numbering owned-first, halo plans).

Reference: Palace partitions the elements of the serial mesh with METIS (utils/geodata.cpp:266-323, :3587-3596) and hands the
parts to MFEM's ParMesh; each rank's local (L-) vector then holds every dof of its elements, a dof shared by several ranks has
one owner, and `y = P^T A_local P x` (linalg/rap.cpp:195-234).  METIS is an external package outside /root/reference; here the
partitioner is recursive coordinate bisection (what SURVEY.md 8(e) proposes in its place) -- any map element -> rank works with
the views below, which replace the slab-only plans of partition.py for unstructured meshes (tetrahedra included).

Every rank builds the SAME global space and partition from the serial mesh (as the reference does before it distributes the
mesh), so the halo plans need no communication: both sides of a rank pair enumerate their shared dofs by global dof number.
Ownership: the lowest rank among the sharers (MFEM's group master)."""
from __future__ import annotations

import numpy as np


def rcb(centroids: np.ndarray, nparts: int) -> np.ndarray:
    """part[e] in [0, nparts): recursive coordinate bisection.  A box is cut along its longest extent so that the two halves
    hold elements in the proportion of the parts they are to be split into (nparts need not be a power of two); ties are
    broken by element number, so the result is deterministic."""
    centroids = np.asarray(centroids, dtype=np.float64)
    ne = centroids.shape[0]
    part = np.zeros(ne, dtype=np.int32)

    def split(ids, first, count):
        if count == 1 or ids.size == 0:
            part[ids] = first
            return
        c = centroids[ids]
        axis = int(np.argmax(c.max(axis=0) - c.min(axis=0))) if ids.size else 0
        left = count // 2
        n_left = int(round(ids.size * left / count))
        order = np.lexsort((ids, c[:, axis]))  # by coordinate, then by element number
        split(ids[order[:n_left]], first, left)
        split(ids[order[n_left:]], first + left, count - left)

    split(np.arange(ne, dtype=np.int64), 0, int(nparts))
    return part


def _elem_dofs(space):
    """[ne, P] global dof numbers of a space (tetrahedral spaces: `offsets`, hexahedral ones: `elem_dof_lex`)."""
    return np.asarray(space.offsets if hasattr(space, "offsets") else space.elem_dof_lex, dtype=np.int64)


class PartitionedSpace:
    """Rank `rank`'s view of `space` (global: every element, global dof numbers) under the element partition `part`.

    owner      [space.ndofs] owning rank of every global dof
    elems      global numbers of the rank's elements (increasing)
    l2g        [ndofs] global dof of local dof: owned dofs first (increasing global number), then the ghosts grouped by owner
               rank (increasing), each group by global number -- a T-vector is a prefix of the L-vector
    n_true, ndofs (= n_local)
    offsets / elem_dof_lex, orients / curl_orients / elem_sign_lex: the rank's rows, dofs renumbered
    nbr, send, recv: the halo plan (send[k]: owned dofs neighbour nbr[k] holds as ghosts, recv[k]: ghosts it owns), both in
               increasing global dof number, which is the order the neighbour uses for the matching list
    """

    def __init__(self, space, part: np.ndarray, rank: int, world: int):
        self.space, self.rank, self.world = space, int(rank), int(world)
        self.p = space.p
        eg = _elem_dofs(space)
        part = np.asarray(part)
        assert part.shape[0] == eg.shape[0] and part.min() >= 0 and part.max() < world
        n_glob = int(space.ndofs)
        # owner of every global dof: the lowest rank that has it
        owner = np.full(n_glob, world, dtype=np.int64)
        np.minimum.at(owner, eg.ravel(), np.repeat(part.astype(np.int64), eg.shape[1]))
        self.owner = owner  # [n_glob] owning rank of every global dof
        self.elems = np.nonzero(part == rank)[0]
        mine = np.unique(eg[self.elems].ravel())
        owned = mine[owner[mine] == rank]
        ghosts = mine[owner[mine] != rank]
        ghosts = ghosts[np.lexsort((ghosts, owner[ghosts]))]
        self.l2g = np.concatenate([owned, ghosts]).astype(np.int64)
        self.n_true, self.ndofs = int(owned.size), int(self.l2g.size)
        g2l = np.full(n_glob, -1, dtype=np.int64)
        g2l[self.l2g] = np.arange(self.ndofs)
        self._g2l = g2l
        loc = g2l[eg[self.elems]].astype(np.int32)
        if hasattr(space, "offsets"):
            self.offsets = loc
            self.curl_orients = None if getattr(space, "curl_orients", None) is None else space.curl_orients[self.elems]
            self.orients = None if getattr(space, "orients", None) is None else space.orients[self.elems]
            self.diagonal_transform = getattr(space, "diagonal_transform", True)
            self.elem = getattr(space, "elem", None)
            self.P = loc.shape[1]
        else:
            self.elem_dof_lex = loc
            if hasattr(space, "elem_sign_lex"):
                self.elem_sign_lex = space.elem_sign_lex[self.elems]
        # halo plan.  Which ranks hold a dof: the parts of the elements that contain it.
        self.nbr, self.send, self.recv = [], [], []
        gh_owner = owner[ghosts]
        # owned dofs other ranks hold: pairs (dof, rank) over all elements, restricted to my owned dofs and foreign parts
        flat_d = eg.ravel()
        flat_p = np.repeat(part.astype(np.int64), eg.shape[1])
        sel = (owner[flat_d] == rank) & (flat_p != rank)
        pairs = np.unique(np.stack([flat_p[sel], flat_d[sel]], axis=1), axis=0) if sel.any() else np.zeros((0, 2), np.int64)
        for q in sorted(set(np.unique(gh_owner).tolist()) | set(np.unique(pairs[:, 0]).tolist())):
            s = pairs[pairs[:, 0] == q, 1]  # increasing global number (np.unique sorts rows)
            r = ghosts[gh_owner == q]       # increasing global number within the owner's group
            self.nbr.append(int(q))
            self.send.append(g2l[s].astype(np.int32))
            self.recv.append(g2l[r].astype(np.int32))

    def restriction(self, interp_range=False):
        """The rank's rows of space.restriction() with local dof numbers (tetrahedral spaces)."""
        r = dict(self.space.restriction(interp_range=interp_range)) if hasattr(self.space, "restriction") else {}
        out = dict(offsets=self.offsets, lsize=self.ndofs)
        for key in ("orients", "curl_orients"):
            if r.get(key) is not None:
                out[key] = r[key][self.elems]
        return out

    def ess_dofs(self, face_mask=None):
        """Essential TRUE dofs of the rank: the global list restricted to its owned dofs, local numbers."""
        g = np.asarray(self.space.ess_dofs(face_mask) if face_mask is not None else self.space.ess_dofs(), dtype=np.int64)
        l = self._g2l[g]
        return np.sort(l[(l >= 0) & (l < self.n_true)]).astype(np.int32)

    def to_local(self, xg: np.ndarray) -> np.ndarray:
        """The rank's L-vector of a global vector."""
        return np.asarray(xg)[self.l2g]
