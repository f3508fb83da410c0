"""bench.py's `cpw_iso` leg alone (config 3's solver loop on surrogate materials), for rocprofv3 --kernel-trace --stats."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ci = bench.cpw_iso_leg(3)
print(json.dumps({"fgmres": ci["fgmres"]}))
