"""How much E-vector traffic would pre-assembling interface dofs inside clusters of elements save?  

Greedy clustering of the bench mesh by face adjacency (up to `K` elements per cluster); a dof whose copies all lie inside
one cluster becomes exclusive to it (stored straight into y after the in-LDS sum), the others keep one E-vector entry
per (cluster, dof) instead of one per (element, dof).  CPU only: python scripts/cluster_estimate.py [n] [nz] [p] [K]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from palace_amd.fem.fespace import NDHexSpace
from palace_amd.fem.mesh import ogrid_cylinder


def clusters_greedy(mesh, K):
    ne = mesh.ne
    f2e = {}
    for e in range(ne):
        for f in mesh.elem_faces[e]:
            f2e.setdefault(int(f), []).append(e)
    nbr = [[] for _ in range(ne)]
    for es in f2e.values():
        if len(es) == 2:
            nbr[es[0]].append(es[1]), nbr[es[1]].append(es[0])
    cid = np.full(ne, -1, dtype=np.int64)
    nc = 0
    for seed in range(ne):
        if cid[seed] >= 0:
            continue
        members = [seed]
        cid[seed] = nc
        while len(members) < K:
            # the free neighbour sharing the most faces with the cluster
            cand = {}
            for m in members:
                for q in nbr[m]:
                    if cid[q] < 0:
                        cand[q] = cand.get(q, 0) + 1
            if not cand:
                break
            best = max(cand.items(), key=lambda kv: (kv[1], -kv[0]))[0]
            cid[best] = nc
            members.append(best)
        nc += 1
    return cid, nc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nz = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    p = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    mesh = ogrid_cylinder(n, nz)
    nd = NDHexSpace(mesh, p)
    dof = nd.elem_dof_lex
    ne, P = dof.shape
    mult = np.bincount(dof.ravel(), minlength=nd.ndofs)
    excl_now = int((mult[dof] == 1).sum())
    ev_now = ne * P - excl_now
    cid, nc = clusters_greedy(mesh, K)
    sizes = np.bincount(cid)
    # per (cluster, dof) pairs
    pair = np.unique(np.stack([np.repeat(cid, P), dof.ravel()], axis=1), axis=0)
    ncl_of_dof = np.bincount(pair[:, 1], minlength=nd.ndofs)
    excl_cluster = int((ncl_of_dof[pair[:, 1]] == 1).sum())
    ev_cluster = pair.shape[0] - excl_cluster
    print(f"mesh {ne} hexes, ND p={p}: {nd.ndofs} dofs, {ne * P} element-local copies")
    print(f"clusters of <= {K}: {nc} (mean size {sizes.mean():.2f}, full {np.mean(sizes == K):.0%})")
    print(f"E-vector entries  now {ev_now} ({ev_now / ne:.1f}/element)  clustered {ev_cluster} ({ev_cluster / ne:.1f}/element)"
          f"  -> x{ev_cluster / ev_now:.2f}")
    print(f"direct stores     now {excl_now}  clustered {excl_cluster}")
    b_now = ev_now * (8 + 8 + 4) + excl_now * 8
    b_cl = ev_cluster * (8 + 8 + 4) + excl_cluster * 8
    print(f"E^T bytes (write + gather read + index): {b_now / ne:.0f} -> {b_cl / ne:.0f} B/element")


if __name__ == "__main__":
    main()
