"""Pricing experiment for "take the E-vector round trip out of HBM" (round-3 review item 2): how fast is the E^T run gather when
its input is as cache-resident as it can ever be?  HIP-event timings on the bench mesh (10.26M dofs, ND p = 3 curl-curl):
  pair        element kernel + gather back to back (what an apply does)
  elem        the element kernel alone, repeated
  gather hot  the gather alone, repeated: everything it reads (E-vector 108 MB, run tables 45 MB) was read a moment ago --
              the 256 MB Infinity Cache holds as much of it as it ever will; an upper bound for ANY scheme that consumes the
              E-vector soon after it is written
  gather cold the gather after 1 GB of unrelated streaming (nothing of its input on the die)
  gather warm the gather right after the element kernel (the normal case), timed alone
If hot is not clearly faster than warm, chunking / fusing the gather behind the element kernel cannot pay.
  python scripts/price_evec_cache.py [dofs]"""
import ctypes as C
import os
import sys

# (pa_debug_apply_phase lives in the ABLATION library only -- `make -C palace_amd/csrc ablate` -- never in the product library)
_abl = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "palace_amd", "lib", "libpalace_amd_ablate.so")
if not os.path.exists(_abl):
    sys.exit("build the ablation library first: make -C palace_amd/csrc ablate")
os.environ.setdefault("PALACE_AMD_LIB", _abl)

sys.path.insert(0, os.getcwd())
import torch

from palace_amd import lib as _lib
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape

dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
ctx = linalg.Context()
n_cross, nz = strong_shape(dofs, 3)
prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=False, shape=(n_cross, nz))
K = prob.curlcurl_par_operator()
n = prob.n_true[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
L = _lib.load()
L.pa_debug_apply_phase.restype = None
L.pa_debug_apply_phase.argtypes = [C.c_int]
big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")  # 1 GiB
big2 = torch.empty(1 << 27, dtype=torch.float64, device="cuda")


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, reps, pre=None):
    """mean time of fn() over reps, each repetition timed alone with events (pre() runs untimed before it)"""
    tot = 0.0
    with torch.cuda.stream(ctx.torch_stream):
        for _ in range(reps):
            if pre:
                pre()
            a, b = ev(), ev()
            a.record()
            fn()
            b.record()
            b.synchronize()
            tot += a.elapsed_time(b)
    return 1e3 * tot / reps


def loop(fn, reps):
    with torch.cuda.stream(ctx.torch_stream):
        for _ in range(20):
            fn()
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def phase(p):
    L.pa_debug_apply_phase(p)


def only(p):
    def f():
        phase(p)
        K.mult(x, y)
        phase(0)
    return f


def flush():
    big2.copy_(big)


for _ in range(200):
    K.mult(x, y)
torch.cuda.synchronize()
res = {"dofs": n}
res["pair_us"] = loop(lambda: K.mult(x, y), 500)
res["elem_loop_us"] = loop(only(1), 500)
res["gather_hot_loop_us"] = loop(only(2), 500)
res["gather_hot_single_us"] = timed(only(2), 50, pre=only(2))
res["gather_warm_single_us"] = timed(only(2), 50, pre=only(1))
res["gather_cold_single_us"] = timed(only(2), 50, pre=flush)
res["elem_single_after_gather_us"] = timed(only(1), 50, pre=only(2))
res["elem_cold_single_us"] = timed(only(1), 50, pre=flush)
print("price_evec_cache:", " ".join(f"{k}={v:.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in res.items()), flush=True)
