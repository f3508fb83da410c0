"""Curved (tri-quadratic, 27-node) hexahedral meshes: the inputs Palace gets from MFEM's `ParMesh`.

This module stands in for the mesh side of the boundary (reference: palace/fem/mesh.cpp:146-209
hands libCEED a mesh-node restriction + H1 basis + node vector + per-element attribute); MFEM is
not vendored in the reference, so the mesh container, the Gmsh reader, uniform refinement and the
synthetic O-grid cylinder (same block topology as reference examples/cylinder/mesh/mesh.jl:75-90)
are written here from scratch.

Conventions (all arrays numpy):
  x           [Nn, 3]   node coordinates
  elem_nodes  [NE, 27]  node ids, lexicographic on the 3x3x3 reference lattice (i fastest)
  attr        [NE]      1-based element attribute
  verts       [NE, 8]   contiguous vertex ids in MFEM hex vertex order
                        v0=(0,0,0) v1=(1,0,0) v2=(1,1,0) v3=(0,1,0) v4..v7 the same at z=1
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# lattice index (i + 3 j + 9 k) of the 8 corners in MFEM vertex order
_CORNER_LATTICE = np.array([0, 2, 8, 6, 18, 20, 26, 24])
# MFEM hex edges (reference element vertex pairs; every one points in a + coordinate direction)
HEX_EDGES = np.array(
    [[0, 1], [1, 2], [3, 2], [0, 3], [4, 5], [5, 6], [7, 6], [4, 7], [0, 4], [1, 5], [2, 6], [3, 7]]
)
# Faces as (c00, c10, c01, c11) corners in the face's (u,v) frame, u,v = the two in-plane
# coordinate axes in increasing axis order.  Order: z=0, y=0, x=1, y=1, x=0, z=1 (MFEM face order).
HEX_FACES_UV = np.array(
    [
        [0, 1, 3, 2],  # z=0: u=x, v=y
        [0, 1, 4, 5],  # y=0: u=x, v=z
        [1, 2, 5, 6],  # x=1: u=y, v=z
        [3, 2, 7, 6],  # y=1: u=x, v=z
        [0, 3, 4, 7],  # x=0: u=y, v=z
        [4, 5, 7, 6],  # z=1: u=x, v=y
    ]
)
# (normal axis, side, u axis, v axis) per face
HEX_FACE_AXES = [(2, 0, 0, 1), (1, 0, 0, 2), (0, 1, 1, 2), (1, 1, 0, 2), (0, 0, 1, 2), (2, 1, 0, 1)]


def _q2_1d(t):
    """Quadratic Lagrange basis on nodes {0, 1/2, 1} at points t: [len(t), 3] values, derivs."""
    t = np.asarray(t, dtype=np.float64)
    B = np.stack([2 * (t - 0.5) * (t - 1), -4 * t * (t - 1), 2 * t * (t - 0.5)], axis=-1)
    G = np.stack([4 * t - 3, -8 * t + 4, 4 * t - 1], axis=-1)
    return B, G


@dataclass
class HexMesh:
    x: np.ndarray
    elem_nodes: np.ndarray
    attr: np.ndarray
    bdr_faces: np.ndarray | None = None  # [NB, 4] node ids of boundary quads (any order)
    bdr_attr: np.ndarray | None = None
    _topo: dict = field(default_factory=dict, repr=False)

    @property
    def ne(self) -> int:
        return self.elem_nodes.shape[0]

    # ---- topology -------------------------------------------------------------------------
    def _build_topology(self):
        if self._topo:
            return
        corner_nodes = self.elem_nodes[:, _CORNER_LATTICE]
        vid, inv = np.unique(corner_nodes, return_inverse=True)
        verts = inv.reshape(corner_nodes.shape).astype(np.int64)
        nv = vid.size
        # edges
        ev = verts[:, HEX_EDGES]  # [NE, 12, 2]
        lo, hi = ev.min(axis=2), ev.max(axis=2)
        ekey = lo * nv + hi
        ukey, efirst, einv = np.unique(ekey, return_index=True, return_inverse=True)
        edge_verts = np.stack([lo.ravel()[efirst], hi.ravel()[efirst]], axis=1)
        # faces: keyed by their three smallest vertices
        fv = np.sort(verts[:, HEX_FACES_UV], axis=2)  # [NE, 6, 4]
        if nv >= 2_000_000:
            raise ValueError("mesh too large for the int64 face key")
        fkey = (fv[..., 0] * nv + fv[..., 1]) * nv + fv[..., 2]
        ufkey, ffirst, finv, fcount = np.unique(fkey, return_index=True, return_inverse=True, return_counts=True)
        face_verts = fv.reshape(-1, 4)[ffirst]
        self._topo = dict(
            nv=nv,
            vert_nodes=vid,
            verts=verts,
            nedges=ukey.size,
            edge_verts=edge_verts,
            face_verts=face_verts,
            elem_edges=einv.reshape(self.ne, 12),
            nfaces=ufkey.size,
            elem_faces=finv.reshape(self.ne, 6),
            face_count=fcount,
        )

    @property
    def verts(self):
        self._build_topology()
        return self._topo["verts"]

    @property
    def nv(self):
        self._build_topology()
        return self._topo["nv"]

    @property
    def nedges(self):
        self._build_topology()
        return self._topo["nedges"]

    @property
    def nfaces(self):
        self._build_topology()
        return self._topo["nfaces"]

    @property
    def edge_verts(self):
        """[nedges, 2] vertex ids (ascending) of every edge."""
        self._build_topology()
        return self._topo["edge_verts"]

    @property
    def face_verts(self):
        """[nfaces, 4] vertex ids (ascending) of every face."""
        self._build_topology()
        return self._topo["face_verts"]

    @property
    def vert_coords(self):
        """[nv, 3] coordinates of the vertices."""
        self._build_topology()
        return self.x[self._topo["vert_nodes"]]

    @property
    def elem_edges(self):
        self._build_topology()
        return self._topo["elem_edges"]

    @property
    def elem_faces(self):
        self._build_topology()
        return self._topo["elem_faces"]

    @property
    def boundary_face_mask(self):
        """[nfaces] bool: faces with a single adjacent element."""
        self._build_topology()
        return self._topo["face_count"] == 1

    def check(self):
        """Consistency: Q2 node count = V+E+F+C and positive Jacobians at element centres."""
        self._build_topology()
        expect = self.nv + self.nedges + self.nfaces + self.ne
        if np.unique(self.elem_nodes).size != expect:
            raise ValueError(
                f"node merge failed: {np.unique(self.elem_nodes).size} nodes, expected {expect}"
            )
        J = self.jacobian_at(np.array([[0.5, 0.5, 0.5]]))
        if not np.all(np.linalg.det(J[:, 0]) > 0):
            raise ValueError("inverted element")

    # ---- geometry -------------------------------------------------------------------------
    def elem_coords(self) -> np.ndarray:
        """[NE, 27, 3] node coordinates per element (lattice order)."""
        return self.x[self.elem_nodes]

    def jacobian_at(self, pts: np.ndarray) -> np.ndarray:
        """J[e, q, i, j] = d x_i / d xi_j at reference points pts [nq, 3]."""
        Bx, Gx = _q2_1d(pts[:, 0])
        By, Gy = _q2_1d(pts[:, 1])
        Bz, Gz = _q2_1d(pts[:, 2])
        X = self.elem_coords().reshape(self.ne, 3, 3, 3, 3)  # [e, k, j, i, comp]
        J = np.empty((self.ne, pts.shape[0], 3, 3))
        J[..., 0] = np.einsum("qk,qj,qi,ekjic->eqc", Bz, By, Gx, X)
        J[..., 1] = np.einsum("qk,qj,qi,ekjic->eqc", Bz, Gy, Bx, X)
        J[..., 2] = np.einsum("qk,qj,qi,ekjic->eqc", Gz, By, Bx, X)
        return J

    def bounding_box(self):
        return self.x.min(axis=0), self.x.max(axis=0)


# ---- Gmsh 2.2 reader -----------------------------------------------------------------------

def _gmsh_hex27_to_lattice() -> np.ndarray:
    """perm[l] = position in Gmsh's 27-node ordering of lattice node l = i + 3j + 9k."""
    c = np.array(
        [[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0], [0, 0, 2], [2, 0, 2], [2, 2, 2], [0, 2, 2]]
    )
    gedges = [(0, 1), (0, 3), (0, 4), (1, 2), (1, 5), (2, 3), (2, 6), (3, 7), (4, 5), (4, 7), (5, 6), (6, 7)]
    gfaces = [(0, 3, 2, 1), (0, 1, 5, 4), (0, 4, 7, 3), (1, 2, 6, 5), (2, 3, 7, 6), (4, 5, 6, 7)]
    pos = [tuple(v) for v in c]
    pos += [tuple((c[a] + c[b]) // 2) for a, b in gedges]
    pos += [tuple(c[list(f)].sum(axis=0) // 4) for f in gfaces]
    pos += [(1, 1, 1)]
    perm = np.empty(27, dtype=np.int64)
    for g, (i, j, k) in enumerate(pos):
        perm[i + 3 * j + 9 * k] = g
    return perm


def read_gmsh22(path: str) -> HexMesh:
    """Read a Gmsh 2.2 (binary or ASCII) file with 27-node hexes (type 12) and 9-node quads (10).

    The reference's own input for the cylinder example is such a file
    (examples/cylinder/mesh/cylinder_hex.msh: 801 nodes, 80 hex27 + 72 quad9).
    """
    data = open(path, "rb").read()

    def section(name):
        a = data.index(b"$" + name + b"\n") + len(name) + 2
        b = data.index(b"$End" + name)
        return a, b

    a, _ = section(b"MeshFormat")
    hdr = data[a : data.index(b"\n", a)].split()
    binary = int(hdr[1]) == 1
    a, b = section(b"Nodes")
    nl = data.index(b"\n", a)
    nn = int(data[a:nl])
    ids = np.empty(nn, dtype=np.int64)
    xyz = np.empty((nn, 3))
    if binary:
        off = nl + 1
        for n in range(nn):
            ids[n], xyz[n, 0], xyz[n, 1], xyz[n, 2] = struct.unpack_from("<iddd", data, off)
            off += 28
    else:
        rows = data[nl + 1 : b].split()
        for n in range(nn):
            ids[n] = int(rows[4 * n])
            xyz[n] = [float(v) for v in rows[4 * n + 1 : 4 * n + 4]]
    idmap = np.full(ids.max() + 1, -1, dtype=np.int64)
    idmap[ids] = np.arange(nn)

    a, b = section(b"Elements")
    nl = data.index(b"\n", a)
    nelem = int(data[a:nl])
    hexes, hattr, quads, qattr = [], [], [], []
    nnodes = {12: 27, 10: 9, 5: 8, 3: 4, 15: 1, 1: 2, 8: 3}
    if binary:
        off = nl + 1
        done = 0
        while done < nelem:
            etype, nfollow, ntags = struct.unpack_from("<iii", data, off)
            off += 12
            npe = nnodes[etype]
            for _ in range(nfollow):
                rec = struct.unpack_from("<" + "i" * (1 + ntags + npe), data, off)
                off += 4 * (1 + ntags + npe)
                phys = rec[1] if ntags > 0 else 0
                nodes = rec[1 + ntags :]
                if etype == 12:
                    hexes.append(nodes)
                    hattr.append(phys)
                elif etype == 10:
                    quads.append(nodes)
                    qattr.append(phys)
            done += nfollow
    else:
        for line in data[nl + 1 : b].splitlines():
            rec = [int(v) for v in line.split()]
            if not rec:
                continue
            etype, ntags = rec[1], rec[2]
            nodes = rec[3 + ntags :]
            phys = rec[3] if ntags > 0 else 0
            if etype == 12:
                hexes.append(nodes)
                hattr.append(phys)
            elif etype == 10:
                quads.append(nodes)
                qattr.append(phys)
    if not hexes:
        raise ValueError("no 27-node hexahedra in " + path)
    g = idmap[np.array(hexes, dtype=np.int64)]
    elem_nodes = g[:, _gmsh_hex27_to_lattice()]
    bdr = idmap[np.array(quads, dtype=np.int64)][:, :4] if quads else None
    mesh = HexMesh(
        x=xyz,
        elem_nodes=elem_nodes,
        attr=np.array(hattr, dtype=np.int32),
        bdr_faces=bdr,
        bdr_attr=np.array(qattr, dtype=np.int32) if quads else None,
    )
    # drop nodes that no hex references (none expected), keep ids otherwise
    mesh.check()
    return mesh


# ---- merging helper ------------------------------------------------------------------------

def _merge_points(pts: np.ndarray, tol: float):
    """Merge coincident points: returns (unique_pts, inverse).  Points are snapped to a lattice of
    pitch `tol` (three 21-bit integer coordinates packed into one int64 key); a post-check on a
    half-shifted lattice catches the rare point that straddles a cell boundary and retries with a
    different offset."""
    lo = pts.min(axis=0)
    span = float(np.max(pts.max(axis=0) - lo))
    if span / tol >= 2**21 - 2:
        raise ValueError("merge tolerance too small for the 21-bit lattice")

    def keys(shift):
        q = np.floor((pts - lo) / tol + shift).astype(np.int64)
        return (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]

    for shift in (0.137, 0.379, 0.613, 0.859):
        _, first, inv = np.unique(keys(shift), return_index=True, return_inverse=True)
        upts = pts[first]
        if np.max(np.abs(upts[inv] - pts)) < 0.1 * tol:
            k2 = np.unique(keys(shift + 0.5)[first])
            if k2.size == first.size:
                return upts, inv
    raise RuntimeError("point merge failed")


# ---- uniform refinement --------------------------------------------------------------------

def refine_uniform(mesh: HexMesh) -> HexMesh:
    """Split every hex27 into 8, new nodes by evaluating the parent's tri-quadratic map (what
    MFEM's `UniformRefinement` does for a nodal mesh: the shape is unchanged)."""
    t = np.linspace(0.0, 1.0, 5)
    B, _ = _q2_1d(t)  # [5, 3]
    X = mesh.elem_coords().reshape(mesh.ne, 3, 3, 3, 3)
    fine = np.einsum("ck,bj,ai,ekjid->ecbad", B, B, B, X)  # [e, 5(z), 5(y), 5(x), 3]
    children = []
    for c in range(2):
        for b in range(2):
            for a in range(2):
                blk = fine[:, 2 * c : 2 * c + 3, 2 * b : 2 * b + 3, 2 * a : 2 * a + 3, :]
                children.append(blk.reshape(mesh.ne, 27, 3))
    pts = np.stack(children, axis=1).reshape(-1, 3)  # [NE*8*27, 3]
    lo, hi = mesh.bounding_box()
    upts, inv = _merge_points(pts, 1e-6 * float(np.max(hi - lo)))
    out = HexMesh(x=upts, elem_nodes=inv.reshape(mesh.ne * 8, 27), attr=np.repeat(mesh.attr, 8))
    out.check()
    return out


# ---- synthetic O-grid cylinder -------------------------------------------------------------

def ogrid_cylinder(n: int, nz: int, radius: float = 2.74, height: float | None = None,
                   m: int | None = None) -> HexMesh:
    """5-block O-grid cylinder of tri-quadratic hexes: a central square (corners on the axes at
    0.4*sqrt(2)*radius, i.e. the reference's 0.8r square rotated by 45 degrees) with n x n
    elements per layer and four outer blocks with m (radial) x n elements, nz layers.
    NE = (n^2 + 4 n m) nz.  Nodes sit on the exact geometry (arc mid-nodes on the circle).
    """
    if height is None:
        height = 2.0 * radius
    if m is None:
        m = n
    c = 0.4 * np.sqrt(2.0) * radius
    corners = np.array([[c, 0.0], [0.0, c], [-c, 0.0], [0.0, -c]])
    zs = np.linspace(0.0, height, 2 * nz + 1)
    blocks = []
    # central block: bilinear map of the square, local x from corner 3 -> 0, local y from 3 -> 2
    s = np.linspace(0.0, 1.0, 2 * n + 1)
    S, T = np.meshgrid(s, s, indexing="xy")  # S varies along x (fastest index)
    P = (
        (1 - S)[..., None] * (1 - T)[..., None] * corners[3]
        + S[..., None] * (1 - T)[..., None] * corners[0]
        + S[..., None] * T[..., None] * corners[1]
        + (1 - S)[..., None] * T[..., None] * corners[2]
    )  # [2n+1 (y), 2n+1 (x), 2]
    blocks.append(P)
    # outer blocks: local x radial (eta), local y angular (xi)
    eta = np.linspace(0.0, 1.0, 2 * m + 1)
    xi = np.linspace(0.0, 1.0, 2 * n + 1)
    for k in range(4):
        a, b = corners[k], corners[(k + 1) % 4]
        side = (1 - xi)[:, None] * a + xi[:, None] * b  # [2n+1, 2]
        th = 0.5 * np.pi * (k + xi)
        arc = radius * np.stack([np.cos(th), np.sin(th)], axis=1)
        P = (1 - eta)[None, :, None] * side[:, None, :] + eta[None, :, None] * arc[:, None, :]
        blocks.append(P)  # [2n+1 (y = xi), 2m+1 (x = eta), 2]
    pts_list, conn_list = [], []
    offset = 0
    for P in blocks:
        ny2, nx2 = P.shape[0], P.shape[1]
        nzz = zs.size
        pts = np.empty((nzz, ny2, nx2, 3))
        pts[..., :2] = P[None]
        pts[..., 2] = zs[:, None, None]
        nid = (offset + np.arange(nzz * ny2 * nx2)).reshape(nzz, ny2, nx2)
        ex, ey, ez = (nx2 - 1) // 2, (ny2 - 1) // 2, nz
        K, Jx, I = np.meshgrid(np.arange(ez), np.arange(ey), np.arange(ex), indexing="ij")
        loc = np.arange(3)
        kk = (2 * K[..., None, None, None] + loc[:, None, None])
        jj = (2 * Jx[..., None, None, None] + loc[None, :, None])
        ii = (2 * I[..., None, None, None] + loc[None, None, :])
        conn = nid[kk, jj, ii].reshape(-1, 27)  # lattice order: i fastest, then j, then k
        pts_list.append(pts.reshape(-1, 3))
        conn_list.append(conn)
        offset += nzz * ny2 * nx2
    pts = np.concatenate(pts_list)
    conn = np.concatenate(conn_list)
    upts, inv = _merge_points(pts, 1e-6 * max(2.0 * radius, height))
    mesh = HexMesh(x=upts, elem_nodes=inv[conn], attr=np.ones(conn.shape[0], dtype=np.int32))
    mesh.check()
    return mesh


def cylinder_for_dofs(target_dofs: float, p: int) -> HexMesh:
    """O-grid cylinder whose order-p Nedelec space has about `target_dofs` unknowns
    (about 3 p^3 unique dofs per hex).  Aspect: height = diameter, so nz ~ 1.2 n keeps the
    elements roughly isotropic."""
    ne = target_dofs / (3.0 * p**3)
    n = max(1, int(round((ne / (5 * 1.2)) ** (1.0 / 3.0))))
    nz = max(1, int(round(ne / (5 * n * n))))
    return ogrid_cylinder(n, nz)
