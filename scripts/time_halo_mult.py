"""Per-rank cost structure of the strong-scaling apply on ONE GPU: the slab a rank gets at --gpus N (N = env NRANKS, default 8)
with its two halo exchanges redirected to the rank itself (RCCL send / receive to self: the launches, packs, stream joins and
copies of a real multi-rank ParOperator::Mult without the xGMI transfer).  PALACE_AMD_OVERLAP=0/1 toggles the halo stream."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
N = int(os.environ.get("NRANKS", "8")); reps = int(os.environ.get("REPS", "500"))
ctx = linalg.Context()
ctx.init_comm_single()
n, nz = strong_shape(10e6, 3)
prob = SlabProblem(ctx, 1, N, 3, 0, levels=False, shape=(n, nz // N), device=False)  # an interior slab: two neighbours
prob.world = 2  # (halos are built)
for s in prob.spaces:  # both neighbours become the rank itself: what it sends up it receives as its own bottom ghosts
    send = np.concatenate(s.send).astype(np.int32); recv = np.concatenate(s.recv).astype(np.int32)
    assert send.size == recv.size
    s.nbr, s.send, s.recv = [0], [send], [recv]
prob._device_setup()
A = prob.curlcurl_par_operator()
nt = prob.n_true[-1]
x = torch.rand(nt, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
for _ in range(50): A.mult(x, y)
with torch.cuda.stream(ctx.torch_stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): A.mult(x, y)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
loc = prob.local_curlcurl
lx = torch.rand(prob.n_local[-1], dtype=torch.float64, device="cuda"); ly = torch.empty_like(lx)
for _ in range(50): loc.mult(lx, ly)
with torch.cuda.stream(ctx.torch_stream):
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): loc.mult(lx, ly)
    e1.record(); torch.cuda.synchronize()
ml = e0.elapsed_time(e1) / reps
print(f"slab of 1/{N}: {prob.mesh.ne} elements, {nt} true dofs, halo {prob.spaces[-1].send[0].size} dofs each way: "
      f"ParOperator::Mult {ms*1e3:.1f} us, local apply alone {ml*1e3:.1f} us, ideal (1-GPU time / {N}) {178.0/N:.1f} us")
