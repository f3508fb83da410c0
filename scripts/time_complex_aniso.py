"""ComplexParOperator::Mult with anisotropic materials on the bench mesh (hexahedra, ND p = 3, 10.26M complex dofs): the one-pass
packed-D complex kernel against the four real applies of linalg/operator.cpp:98-134 (PALACE_AMD_COMPLEX_FUSED=0); PARITY=1 adds
the C oracle's four real applies at this size.  Also the real anisotropic K + M apply (packed 12 doubles per point) alone.
Prints one JSON line.
  python scripts/time_complex_aniso.py; PALACE_AMD_COMPLEX_FUSED=0 python scripts/time_complex_aniso.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.partition import SlabProblem  # noqa: E402

ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10.0e6")), levels=False)
out = bench.complex_leg(ctx, prob, reps=50, parity=os.environ.get("PARITY", "0") == "1", aniso=True)
out["fused_env"] = os.environ.get("PALACE_AMD_COMPLEX_FUSED", "1")
# the real anisotropic K + M operator alone (one of the four applies of the unfused form)
nd = prob.spaces[-1]
eps = np.diag([9.3, 9.3, 11.5])
Ar = ceed.curlcurlmass_operator(prob.geom, nd, ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[-0.3 * eps]), ceed.coefficient_context(3))
A = linalg.ParOperator(ctx, Ar, prob.ess[-1], linalg.DIAG_ONE)
x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for _ in range(20):
    A.mult(x, y)
with torch.cuda.stream(ctx.torch_stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        A.mult(x, y)
    e1.record()
    torch.cuda.synchronize()
out["real_aniso_curlcurl_mass_ms"] = e0.elapsed_time(e1) / 100
out["real_streams"] = int(ceed._lib.load().pa_op_streams(Ar.handle))
print(json.dumps(out))
