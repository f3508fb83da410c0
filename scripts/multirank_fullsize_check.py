"""Bench-size check of the multi-rank paths on one GPU: the strong-scaling cylinder (10.26M dofs) solved with 1 and with 8 ranks
(threads, in-process communicator) -- fixed 10 PCG + p-multigrid iterations, auxiliary-space smoother; prints the global
quantities of both runs."""
import os, sys, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape

def rank_main(group, rank, world, out):
    torch.cuda.set_device(0)
    ctx = linalg.Context()
    if world > 1:
        ctx.init_comm_local(group, rank)
    n, nz = strong_shape(10e6, 3)
    prob = SlabProblem(ctx, rank, world, 3, 0, shape=(n, nz // world))
    K, b, x = prob.pcg_gmg_solver(max_it=10, rel_tol=0.0, hiptmair=True, coarse="cg")
    K.mult(b, x)
    st = K.stats()
    A = prob._keep[-1][1][-1]
    z = torch.zeros_like(x)
    A.mult(b, z)
    out[rank] = dict(n=int(prob.n_true[-1]), it=st["iterations"], res=st["final_res"], bb=ctx.dot(b, b), bAb=ctx.dot(b, z), xx=ctx.dot(x, x))
    ctx.synchronize()

def run(world):
    group = linalg.LocalGroup(world) if world > 1 else None
    out = [None] * world
    th = [threading.Thread(target=rank_main, args=(group, r, world, out), daemon=True) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=900) for t in th]
    assert all(o is not None for o in out), "a rank failed"
    r = dict(out[0]); r["n"] = sum(o["n"] for o in out)
    return r

one = run(1); print("1 rank :", one, flush=True)
many = run(int(os.environ.get("NRANKS", "8"))); print("N ranks:", many, flush=True)
for k in ("bb", "bAb", "xx", "res"):
    print(k, abs(many[k] - one[k]) / abs(one[k]))
assert many["n"] == one["n"]
