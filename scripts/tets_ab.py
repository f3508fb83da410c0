"""bench.py's `tets` leg (config 3's element: curl-curl and K + M applies on 279 936 order-3 Nedelec tetrahedra) under the environment's
PALACE_AMD_DENSE_ELAYOUT / PALACE_AMD_DENSE_GATHER_GROUP; one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = bench.tets_leg(3, int(os.environ.get("TET_N", "36")))
pick = lambda d: {k: (round(v, 5) if isinstance(v, float) else v) for k, v in d.items() if k in ("ms", "hbm_frac", "mfma_frac", "TFLOPs", "dof_per_s")}
print(json.dumps({"elayout": os.environ.get("PALACE_AMD_DENSE_ELAYOUT", "default"), "group": os.environ.get("PALACE_AMD_DENSE_GATHER_GROUP", "default"),
                  **{k: pick(v) for k, v in t.items() if isinstance(v, dict) and "ms" in v}}))
