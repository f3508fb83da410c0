"""The [Parallel] cases of the reference's orthogonalisation unit test (test/unit/test-orthog.cpp:123-268) on two ranks
(gloo): rank-wise orthogonal basis vectors ("Real 1", "Complex 1") and the known-answer coefficients of "Real 2",
H[0] = size (size - 1) / (2 |v0|), H[1] = size (size + 1) / (2 |v1|) with |v| = sqrt(size)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import palace_oracle as po

        def gsum(a):
            a = np.asarray(a)
            if np.iscomplexobj(a):
                t = torch.from_numpy(np.stack([a.real, a.imag]).copy())
                dist.all_reduce(t)
                return t.numpy()[0] + 1j * t.numpy()[1]
            t = torch.from_numpy(a.astype(np.float64).copy())
            dist.all_reduce(t)
            return t.numpy()

        def gdot(x, y):  # Dot(x, y) = y^H x, globally
            return gsum(np.array([np.vdot(y, x)]))[0]

        res = {}
        for kind in ("MGS", "CGS", "CGS2"):
            # "Real 1": V[r] = e_r on rank r only
            V = [np.zeros(world + 1) for _ in range(world)]
            V[rank][rank] = 1.0
            w = np.random.default_rng(10 + rank).uniform(-1, 1, world + 1)
            H, w2 = po.orthogonalize_column(kind, V, w, world, global_sum=gsum)
            res[kind, "real1"] = max(abs(w2[rank]), max(abs(gdot(w2, V[i])) for i in range(world)))
            # "Real 2"
            V = [np.array([1.0, 0, 0, 0]), np.array([0.0, 1, 0, 0])]
            n0 = np.sqrt(gdot(V[0], V[0]).real)
            V[0] = V[0] / n0
            _, v1 = po.orthogonalize_column(kind, V, V[1], 1, global_sum=gsum)
            exact = bool(np.array_equal(v1, [0.0, 1.0, 0.0, 0.0]))
            n1 = np.sqrt(gdot(v1, v1).real)
            V[1] = v1 / n1
            w = rank + np.arange(4.0)
            H, w2 = po.orthogonalize_column(kind, V, w, 2, global_sum=gsum)
            res[kind, "real2"] = (exact, abs(gdot(w2, V[0])), abs(gdot(w2, V[1])), w2[2] - (rank + 2.0), w2[3] - (rank + 3.0),
                                  H[0] - world * (world - 1.0) / (2 * n0), H[1] - world * (world + 1.0) / (2 * n1),
                                  n0 - np.sqrt(world))
            # "Complex 1"
            V = [np.zeros(world + 1, dtype=complex) for _ in range(world)]
            th = 2 * np.pi * rank / world
            V[rank][rank] = np.cos(th) + 1j * np.sin(th)
            rng = np.random.default_rng(20 + rank)
            w = rng.uniform(-1, 1, world + 1) + 1j * rng.uniform(-1, 1, world + 1)
            H, w2 = po.orthogonalize_column(kind, V, w, world, global_sum=gsum)
            res[kind, "complex1"] = max(abs(w2[rank]), max(abs(gdot(w2, V[i])) for i in range(world)))
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_orthogonalize_column_two_ranks():
    world = 2
    port = 29650 + os.getpid() % 200
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for rank in range(world):
        for kind in ("MGS", "CGS", "CGS2"):
            assert out[rank][kind, "real1"] < 1e-12
            exact, d0, d1, e2, e3, h0, h1, nn = out[rank][kind, "real2"]
            assert exact and d0 < 1e-12 and d1 < 1e-12 and e2 == 0.0 and e3 == 0.0
            assert abs(h0) < 1e-14 and abs(h1) < 1e-14 and abs(nn) < 1e-15
            assert out[rank][kind, "complex1"] < 1e-12
