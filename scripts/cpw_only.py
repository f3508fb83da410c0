"""bench.py's `cpw` leg alone (config 3 with the reference's materials, ports and absorbing boundary), for rocprofv3 --kernel-trace."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ci = bench.cpw_leg(3)
print(json.dumps({"fgmres": ci["fgmres"]}))
