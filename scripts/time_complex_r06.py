"""One-pass complex applies at the bench size, isotropic (metric form) and anisotropic (packed form): bench.py's complex_leg without the
parity leg, one JSON line.  Environment switches read by the library: PALACE_AMD_CPLX_GATHER2=0 (two run gathers instead of one),
PALACE_AMD_STREAM_AFFINE=0 (no affine batches)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=False, shape=strong_shape(10.0e6, 3))
out = {"iso": bench.complex_leg(ctx, prob, reps=100, parity=False), "aniso": bench.complex_leg(ctx, prob, reps=100, parity=False, aniso=True),
       "env": {k: os.environ.get(k) for k in ("PALACE_AMD_CPLX_GATHER2", "PALACE_AMD_STREAM_AFFINE")}}
print(json.dumps(out))
