"""A / B of the streaming curl-curl kernel with the geometry recomputed from the 27 nodes of an element (PALACE_AMD_STREAM_GEOM=nodes,
PALACE_AMD_GEOMN_VARIANT=w3g1|w3g2|w2g1|w2g2) against the packed q-data form: the headline ParOperator::Mult at the bench size, and
the result against the packed form's (the first run, without the switch, leaves its y in /tmp/geomn_ref.pt).
  python scripts/geomn_ab.py; PALACE_AMD_STREAM_GEOM=nodes python scripts/geomn_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
mode = os.environ.get("PALACE_AMD_STREAM_GEOM", "packed") + ":" + os.environ.get("PALACE_AMD_GEOMN_VARIANT", "-")
dofs = float(os.environ.get("DOFS", "10.0e6"))
ctx = linalg.Context()
n_cross, nz = strong_shape(dofs, 3)
prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=False, shape=(n_cross, nz))
K = prob.curlcurl_par_operator()
n = prob.n_true[-1]
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g); y = torch.empty_like(x)
K.mult(x, y)
torch.cuda.synchronize()
ref = "/tmp/geomn_ref.pt"
if mode.startswith("packed"):
    torch.save(y.cpu(), ref)
    rel = 0.0
else:
    yr = torch.load(ref).cuda()
    rel = float((y - yr).norm() / yr.norm())
with torch.cuda.stream(ctx.torch_stream):
    for _ in range(300): K.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): K.mult(x, y)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 1000
print(f"GEOM {mode}: ParOperator::Mult {ms * 1e3:.1f} us, {n / ms / 1e6:.2f} Gdof/s, rel diff from the packed form {rel:.2e} ({prob.mesh.ne} elements, {n} dofs)", flush=True)
