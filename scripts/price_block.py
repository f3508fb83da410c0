"""Price of assembling the copies of a dof inside a wave's 4 elements / a workgroup's 8 elements on chip before the E-vector store
(PALACE_AMD_PRICE_BLOCK=4 | 8: wrong results, right bytes -- see build_stream): the headline ParOperator::Mult at the bench size.
  for g in 0 4 8; do PALACE_AMD_PRICE_BLOCK=$g python scripts/price_block.py; done"""
import os, sys
# (the pricing switches live in the ABLATION library only -- `make -C palace_amd/csrc ablate` -- never in the product library)
_abl = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "palace_amd", "lib", "libpalace_amd_ablate.so")
if not os.path.exists(_abl):
    sys.exit("build the ablation library first: make -C palace_amd/csrc ablate")
os.environ.setdefault("PALACE_AMD_LIB", _abl)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
g = os.environ.get("PALACE_AMD_PRICE_BLOCK", "0")
if g == "0":
    os.environ.pop("PALACE_AMD_PRICE_BLOCK", None)
ctx = linalg.Context()
n_cross, nz = strong_shape(10.0e6, 3)
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=False, shape=(n_cross, nz))
K = prob.curlcurl_par_operator()
n = prob.n_true[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
with torch.cuda.stream(ctx.torch_stream):
    for _ in range(300): K.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): K.mult(x, y)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 1000
print(f"PRICE_BLOCK={g}: ParOperator::Mult {ms * 1e3:.1f} us, {n / ms / 1e6:.2f} Gdof/s ({prob.mesh.ne} elements, {n} dofs)", flush=True)
