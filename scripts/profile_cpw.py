"""Profile target: BASELINE config 3's solver loop on the reference's cpw mesh (bench.py's `cpw` leg, nothing else):
  rocprofv3 --kernel-trace --marker-trace --stats -- python scripts/profile_cpw.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

leg = bench.cpw_iso_leg if os.environ.get("ISO") else bench.cpw_leg  # ISO=1: the surrogate isotropic leg of rounds 3-4
out = leg(int(os.environ.get("ORDER", "3")), int(os.environ.get("REFINE", "1")))
print("cpw:", json.dumps({k: v for k, v in out.items() if k != "workload"}), flush=True)
