"""Per-rank cost structure of the strong-scaling apply on ONE GPU: the slab a rank gets at --gpus N (N = env NRANKS, default 8)
with its two halo exchanges redirected to the rank itself (RCCL send / receive to self: the launches, packs, stream joins and
copies of a real multi-rank ParOperator::Mult without the xGMI transfer).  HALO_MODE=peer (default: the peer transport's
kernels storing into the rank's own mailboxes) | rccl; PALACE_AMD_OVERLAP=0/1 toggles the halo stream."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
N = int(os.environ.get("NRANKS", "8")); reps = int(os.environ.get("REPS", "500"))
ctx = linalg.Context()
MODE = os.environ.get("HALO_MODE", "peer")  # peer: direct stores into the (own) mailboxes; rccl: send / receive groups to self
if MODE == "peer":
    ctx.init_comm_peer_single()
elif MODE != "none":  # none: the same slab as a one-rank problem (no halos at all): what the launch count alone costs at this size
    ctx.init_comm_single()
n, nz = strong_shape(10e6, 3)
PCG = int(os.environ.get("PCG", "0"))  # > 0: also PCG + p-multigrid iterations / s on the slab, halo exchanges on every level
prob = SlabProblem(ctx, 1, N, 3, 0, levels=bool(PCG), shape=(n, nz // N), device=False)  # an interior slab: two neighbours
if MODE == "none":
    prob = SlabProblem(ctx, 0, 1, 3, 0, levels=bool(PCG), shape=(n, nz // N), device=False)
else:
    prob.world = N  # (> 1: halos are built)
for s in (prob.spaces if MODE != "none" else []):  # both neighbours become the rank itself: what it sends up it receives as its own bottom ghosts
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
if MODE == "peer":
    ctx.peer_check()
print(f"[{MODE}] slab of 1/{N}: {prob.mesh.ne} elements, {nt} true dofs, halo {prob.spaces[-1].send[0].size if MODE != 'none' else 0} dofs each way: "
      f"ParOperator::Mult {ms*1e3:.1f} us, local apply alone {ml*1e3:.1f} us, ideal (1-GPU time / {N}) {178.0/N:.1f} us")

if PCG:
    import time
    from palace_amd.fem import partition as _pt
    # the auxiliary H1 spaces get the same self-neighbour plans
    _orig = _pt.SlabH1Space
    class _SelfH1(_orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            send = np.concatenate(self.send).astype(np.int32); recv = np.concatenate(self.recv).astype(np.int32)
            self.nbr, self.send, self.recv = [0], [send], [recv]
    if MODE != "none":
        _pt.SlabH1Space = _SelfH1
    for hip in (False, True):
        solver, b, xs = prob.pcg_gmg_solver(max_it=PCG, hiptmair=hip, coarse="cg" if hip else "chebyshev")
        solver.mult(b, xs); solver.mult(b, xs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = solver.stats()
        try:
            fused = [bool(prob.last_gmg.fused_step(l)) for l in range(1, len(prob.spaces))] if not hip else "-"
        except Exception as exc:  # noqa: BLE001
            fused = f"? ({exc})"
        print(f"[{MODE}] PCG + p-MG ({'hiptmair' if hip else 'chebyshev'}) on the slab with halos: {st['iterations'] / dt:.0f} it/s "
              f"({st['iterations']} iterations, {dt * 1e3:.1f} ms; smoother steps fused per level: {fused})")
        prob._keep.clear()
    if MODE == "peer":
        ctx.peer_check()
