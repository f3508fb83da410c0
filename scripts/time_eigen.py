"""bench.py's `eigenmode` leg alone (BASELINE config 2's shape): python scripts/time_eigen.py [order] [dofs] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dofs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0e6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
print("eigen:", json.dumps(bench.eigen_leg(order, dofs, steps)), flush=True)
