"""Timeline of the streaming kernel (library built with -DPA_STREAM_TRACE, PALACE_AMD_LIB=...): per-phase cycles of wave 0
of the first 64 workgroups over their first batches."""
import ctypes as C
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg, lib
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")), levels=False)
which = os.environ.get("OP", "curl")
if which == "curl":
    op = prob.local_curlcurl
else:
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    op = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, ceed.coefficient_context(3))
n = prob.n_local[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(3):
    op.mult(x, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): op.mult(x, y)
e1.record(); torch.cuda.synchronize()
print(f"{which}: mult {e0.elapsed_time(e1)/10:.4f} ms")
L = lib.load()
NW, NB, NS = 64, 14, 12
buf = (C.c_ulonglong * (NW * NB * NS))()
rc = L.pa_debug_stream_trace(buf, NW * NB * NS)
assert rc > 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NB, NS).astype(np.int64)
names = ["0 top->x staged", "1 stage->q/ticket issued", "2 forward", "3 wait q+ticket", "4 idx issue + D", "5 bwd comp0",
         "6 wait idx + x issue", "7 bwd comp1,2", "8 E^T + stores"]
valid = t[:, :, 9] > 0
nb = valid.sum(1)
print("batches per traced wave: min", nb.min(), "max", nb.max(), "mean", nb.mean())
d = np.diff(t[:, :, :10], axis=2)  # [w, b, 9]
gap = t[:, 1:, 0] - t[:, :-1, 9]  # loop back edge
for k, nm in enumerate(names):
    v = d[:, :, k][valid]
    print(f"{nm:28s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
tot = (t[:, :, 9] - t[:, :, 0])[valid]
print(f"{'whole batch':28s} mean {tot.mean():9.0f}  median {np.median(tot):9.0f}")
vg = valid[:, 1:] & valid[:, :-1]
print(f"{'back edge':28s} mean {gap[vg].mean():9.0f}")
w = 0
print("wave 0 of workgroup 0, per batch:")
for b in range(NB):
    if valid[w, b]:
        print(b, " ".join(f"{v:7d}" for v in d[w, b]))
span = (t[:, :, 9].max(axis=1) - t[:, 0, 0])
print("span of traced waves (cycles): mean", span.mean(), "max", span.max())
