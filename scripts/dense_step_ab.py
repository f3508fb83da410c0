"""The fused smoother step on dense-table blocks (round 6): bench.py's `spheres` (config 4: H1 tetrahedra, PCG + p-multigrid + AMG) and
`cpw_iso` (config 3 with surrogate materials: complex FGMRES, real Hiptmair p-multigrid on both parts) legs with PALACE_AMD_FUSED_STEP as
the environment says; one JSON line.  Run once with =1 and once with =0."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sp = bench.spheres_leg()
ci = bench.cpw_iso_leg(3)
print(json.dumps({"fused_step": os.environ.get("PALACE_AMD_FUSED_STEP", "1"),
                  "spheres_iters_per_s": {k: round(v["iters_per_s"], 1) for k, v in sp.items() if isinstance(v, dict) and "iters_per_s" in v},
                  "cpw_iso_fgmres": {k: ci["fgmres"][k] for k in ("iterations_to_1e-8", "seconds", "iters_per_s") if k in ci.get("fgmres", {})}}))
