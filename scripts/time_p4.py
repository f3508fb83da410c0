"""Event timings of the order-4 operator applies (BASELINE config 5's element) for A/B runs: the five-point streaming
kernel (pa_nd_hex_stream5.hip) against the one-shot kernel, and launch parameters of the former.

  python scripts/time_p4.py [dofs]        env: P4_VARIANTS="default,g1q0,g1q1,g1q2,g2q0,g2q1,g2q2,wg4,wg5"
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem

dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
variants = os.environ.get("P4_VARIANTS", "default,g1q0,g1q1,g1q2,g2q0,g2q1,g2q2,wg4,wg5").split(",")
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 4, dofs, levels=False)
nd, geom = prob.spaces[-1], prob.geom
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
ident = ceed.coefficient_context(3)
print(f"p=4: {prob.mesh.ne} elements, {nd.ndofs} dofs", flush=True)
x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
ENV = {"default": {}, "oneshot": {"PALACE_AMD_STREAM5": "0"},
       "gpos1": {"PALACE_AMD_STREAM5_GPOS": "1"}, "gpos2": {"PALACE_AMD_STREAM5_GPOS": "2"},
       "g1q0": {"PALACE_AMD_STREAM5_GPOS": "1", "PALACE_AMD_STREAM5_QPOS": "0"},
       "g1q1": {"PALACE_AMD_STREAM5_GPOS": "1", "PALACE_AMD_STREAM5_QPOS": "1"},
       "g1q2": {"PALACE_AMD_STREAM5_GPOS": "1", "PALACE_AMD_STREAM5_QPOS": "2"},
       "g2q0": {"PALACE_AMD_STREAM5_GPOS": "2", "PALACE_AMD_STREAM5_QPOS": "0"},
       "g2q1": {"PALACE_AMD_STREAM5_GPOS": "2", "PALACE_AMD_STREAM5_QPOS": "1"},
       "g2q2": {"PALACE_AMD_STREAM5_GPOS": "2", "PALACE_AMD_STREAM5_QPOS": "2"},
       "wg5": {"PALACE_AMD_STREAM_WG": "5"}, "wg6": {"PALACE_AMD_STREAM_WG": "6"},
       "wg1": {"PALACE_AMD_STREAM_WG": "1"}, "wg2": {"PALACE_AMD_STREAM_WG": "2"}, "wg3": {"PALACE_AMD_STREAM_WG": "3"},
       "wg4": {"PALACE_AMD_STREAM_WG": "4"}}
ref = {}
for v in variants:
    for k in ("PALACE_AMD_STREAM5", "PALACE_AMD_STREAM5_GPOS", "PALACE_AMD_STREAM5_QPOS", "PALACE_AMD_STREAM_WG"):
        os.environ.pop(k, None)
    os.environ.update(ENV[v])
    ops = {"curlcurl": ceed.curlcurl_operator(geom, nd, ident), "curlcurl_mass": ceed.curlcurlmass_operator(geom, nd, mass, ident),
           "mass": ceed.ndmass_operator(geom, nd, mass)}
    line = [f"{v:8s}"]
    for name, op in ops.items():
        with torch.cuda.stream(ctx.torch_stream):
            for _ in range(20):
                op.mult(x, y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                op.mult(x, y)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        frac = op.algorithmic_bytes() / ms / 1e6 / 8000.0
        chk = float(torch.linalg.norm(y))
        ref.setdefault(name, chk)
        line.append(f"{name} {ms:.4f} ms ({frac:.3f}) d={abs(chk - ref[name]) / ref[name]:.1e}")
    print("  ".join(line), flush=True)
    del ops
