"""Rehearsal of the multi-GPU bench on one GPU: `bench.py --rehearse N` starts N rank processes on device 0 (gloo + the peer
transport) that run the very code of `bench.py --gpus N` -- strong z-slabs, halo plans, the direct form of ParOperator::Mult,
both PCG legs and the replicated AMS leg, the order-4 and tetrahedral N-rank legs.  The line must be complete, the two forms of
the multi-rank Mult must agree to the bit, and the solves must take the iterations of the one-rank run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--dofs", "3e5", "--steps", "20", "--warmup", "5", "--pre-warm", "5", "--pcg-iters", "5", "--tet-n", "6", "--no-cpu",
         "--no-p4", "--no-tets"]


def _run(extra):
    env = dict(os.environ, PALACE_AMD_PEER_TIMEOUT_S="60")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, cwd=ROOT, env=env, capture_output=True,
                       timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def one_rank():
    return _run(["--gpus", "1"])


@pytest.mark.parametrize("n", [2, 4, 8])
def test_rehearsal_line(n, one_rank):
    # (4 ranks: the headline and the PCG legs only; 8 ranks -- eight processes time-slicing one GPU, every wait of one a time slice of
    # another -- the headline, the transport bring-up and the cross-check of the two forms of Mult: the suite's time.  The full
    # eight-rank line at the bench size is profiles/r04_rehearse8_bench_size.json.)
    extra = {2: [], 4: ["--no-nranks-legs"], 8: ["--no-nranks-legs", "--pcg-iters", "0"]}[n]
    out = _run(["--rehearse", str(n)] + extra)
    assert out["rehearsal"] is True and out["n_gpus"] == n and out["scaling"] == "strong"
    assert out["config"]["global_true_dofs"] == one_rank["config"]["global_true_dofs"]
    halo = out["halo"]
    assert halo["transport"].startswith("peer") and halo["direct_form"] == 1
    assert halo["direct_vs_lvector_rel_err"] == 0.0  # the same numbers summed in the same order
    st = halo["bring_up"]["self_test"]
    assert st["known_answers"] and st["lvector_wrong_values"] == 0 and st["direct_wrong_values"] == 0 and st["direct_graph_wrong_values"] == 0
    assert halo["partition"]["neighbours"] in (1, 2) and halo["partition"]["ghost_dofs"] >= 0
    assert out["value"] > 0 and out["ms_per_step"] > 0
    if n == 8:
        assert out["pcg"] is None and out["n_ranks_legs"] is None
        return
    for leg in ("chebyshev", "hiptmair", "hiptmair_ams", "hiptmair_ams_distributed"):
        e = out["pcg"][leg]
        assert "error" not in e, e
        assert e["converged"], (leg, e)
    # (hiptmair_ams_distributed, round 5: the algebraic level-0 cycle with its solve distributed over the ranks, amg_dist.hpp)
    for leg, ref in (("chebyshev", "chebyshev"), ("hiptmair", "hiptmair"), ("hiptmair_ams", "hiptmair_ams"),
                     ("hiptmair_ams_distributed", "hiptmair_ams")):
        a, b = out["pcg"][leg]["iterations_to_1e-8"], one_rank["pcg"][ref]["iterations_to_1e-8"]
        assert abs(a - b) <= max(2, b // 10), (leg, a, b)
    assert 0.0 < out["pcg"]["hiptmair_ams"]["replicated_level0"]["share_of_iteration"] < 1.0
    assert 0.0 < out["pcg"]["hiptmair_ams_distributed"]["distributed_level0"]["share_of_iteration"] < 1.0
    legs = out["n_ranks_legs"]
    if n == 4:
        assert legs is None
        return
    assert legs["peer_check"] == "ok"
    for leg in ("p4", "tets"):
        assert "error" not in legs[leg], legs[leg]
        assert legs[leg]["direct_form"] == 1 and legs[leg]["dof_per_s"] > 0  # (hexahedra and tetrahedra: split-vector applies)
    assert legs["tets"]["pcg_hiptmair_ams"]["converged"]
