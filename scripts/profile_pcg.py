import os, sys
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")))
K, b, x = prob.pcg_gmg_solver(max_it=int(os.environ.get("ITS", "20")), hiptmair=os.environ.get("HIP", "1") == "1")
K.mult(b, x); torch.cuda.synchronize()
print(K.stats())
