"""Time one Arnoldi column (orthogonalise + norm + normalise) at BASELINE config 3's size: device-resident coefficients
(orthog.hip) against the host-driven form, MGS / CGS / CGS2, real and complex.
  python scripts/time_orthog.py [n] [m]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from palace_amd import linalg

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2180208
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = linalg.Context()
g = torch.Generator(device="cuda").manual_seed(1)
Vr = [torch.randn(n, dtype=torch.float64, device="cuda", generator=g) / np.sqrt(2 * n) for _ in range(m)]
Vi = [torch.randn(n, dtype=torch.float64, device="cuda", generator=g) / np.sqrt(2 * n) for _ in range(m)]
w0r, w0i = torch.randn(n, dtype=torch.float64, device="cuda", generator=g), torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
for cplx in (True, False):
    for kind in ("MGS", "CGS", "CGS2"):
        res = {}
        for dev in (True, False):
            linalg.Context.set_device_orthogonalization(dev)
            ts = []
            for rep in range(4):
                wr, wi = w0r.clone(), w0i.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if cplx:
                    H, hn = ctx.orthonormalize_column_complex(kind, Vr, Vi, wr, wi)
                else:
                    H, hn = ctx.orthonormalize_column(kind, Vr, wr)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[dev] = (min(ts[1:]), hn, np.abs(H).max())
        passes = {"MGS": 4 if res else 5, "CGS": 2, "CGS2": 4}[kind]
        gb = passes * m * n * (16 if cplx else 8) / 1e9
        print(f"{'complex' if cplx else 'real   '} {kind:4s} n={n} m={m}: device {res[True][0] * 1e3:8.3f} ms ({gb / res[True][0]:7.0f} GB/s of basis reads), "
              f"host-driven {res[False][0] * 1e3:8.3f} ms; hn {res[True][1]:.15e} vs {res[False][1]:.15e}", flush=True)
linalg.Context.set_device_orthogonalization(True)
