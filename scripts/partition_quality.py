"""Quality of the 8-way element partitions (METIS is external; recursive coordinate bisection / z-slabs stand in for
it): per rank the elements, true dofs, ghosts, owned dofs sent, neighbours and surface / volume = (ghosts + sent) / true dofs, for the
meshes of the bench legs.  CPU only.   python scripts/partition_quality.py [world]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from palace_amd.fem import tet
from palace_amd.fem.partition import SlabNDSpace, strong_shape
from palace_amd.fem.mesh import ogrid_cylinder
from palace_amd.fem.rcb import PartitionedSpace, rcb

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(name, views):
    rows = []
    for s in views:
        ns = int(sum(len(q) for q in s.send))
        nr = int(sum(len(q) for q in s.recv))
        ne = int(len(s.elems)) if hasattr(s, "elems") else int(s.mesh.ne)
        rows.append((ne, int(s.n_true), nr, ns, len(s.nbr), (ns + nr) / max(1, s.n_true)))
    a = np.array(rows, dtype=np.float64)
    print(f"{name}: {len(views)} ranks")
    print(f"  elements per rank      {a[:, 0].min():.0f} .. {a[:, 0].max():.0f}   (imbalance {a[:, 0].max() / a[:, 0].mean():.3f})")
    print(f"  true dofs per rank     {a[:, 1].min():.0f} .. {a[:, 1].max():.0f}   (imbalance {a[:, 1].max() / a[:, 1].mean():.3f})")
    print(f"  ghosts / sent per rank {a[:, 2].min():.0f} .. {a[:, 2].max():.0f} / {a[:, 3].min():.0f} .. {a[:, 3].max():.0f}")
    print(f"  neighbours             {a[:, 4].min():.0f} .. {a[:, 4].max():.0f}")
    print(f"  surface / volume       {a[:, 5].min():.4f} .. {a[:, 5].max():.4f}   (mean {a[:, 5].mean():.4f})", flush=True)


# 1. the strong-scaling bench cylinder, ND p = 3, z-slabs
n, nz = strong_shape(10e6, 3)
h_layer = 2.0 * 2.74 / max(1, round(1.15 * n))
views = []
for r in range(world):
    height = (nz // world) * h_layer
    mesh = ogrid_cylinder(n, nz // world, radius=2.74, height=height)
    mesh.x = mesh.x.copy()
    mesh.x[:, 2] += r * height
    views.append(SlabNDSpace(mesh, 3, r, world, r * height, (r + 1) * height, 2.74))
report(f"bench cylinder ({5 * n * n * nz} hex27, ND p=3), {world} z-slabs", views)

# 2. Kuhn-split cube, ND p = 3, recursive coordinate bisection
mesh = tet.cube_tet_mesh(36)
sp = tet.NDTetSpace(mesh, 3)
part = rcb(mesh.nodes[mesh.elem_nodes[:, :4]].mean(axis=1), world)
report(f"cube of {mesh.ne} tetrahedra, ND p=3, RCB", [PartitionedSpace(sp, part, r, world) for r in range(world)])

# 3. the reference's cpw mesh refined once, ND p = 3, RCB
d = np.load(os.path.join(ROOT, "tests", "golden", "cpw_mesh.npz"))
mesh = tet.refine_uniform(tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"]))
sp = tet.NDTetSpace(mesh, 3)
part = rcb(mesh.nodes[mesh.elem_nodes[:, :4]].mean(axis=1), world)
report(f"examples/cpw mesh refined once ({mesh.ne} tetrahedra), ND p=3, RCB", [PartitionedSpace(sp, part, r, world) for r in range(world)])

# 4. the reference's spheres mesh, H1 p = 2, RCB
d = np.load(os.path.join(ROOT, "tests", "golden", "spheres_mesh.npz"))
en = d["elem_nodes"].astype(np.int64)
used, inv = np.unique(en[:, :4], return_inverse=True)
mesh = tet.TetMesh(d["nodes"][used], inv.reshape(-1, 4), d["attr"])
sp = tet.H1TetSpace(mesh, 2)
part = rcb(mesh.nodes[mesh.elem_nodes[:, :4]].mean(axis=1), world)
report(f"examples/spheres mesh ({mesh.ne} tetrahedra), H1 p=2, RCB", [PartitionedSpace(sp, part, r, world) for r in range(world)])
