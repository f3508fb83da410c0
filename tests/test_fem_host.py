"""Host-side pieces of the C++ front end (palace_amd/csrc/fem.hpp, ksp.hpp) against the numpy restatements the rest of
the suite uses: Gauss-Legendre / Gauss-Lobatto points, Lagrange tables, MaterialPropertyCoefficient bookkeeping and the
coefficient contexts (fem/libceed/coefficient.cpp:51-131), p-coarsening sequences (fem/multigrid.hpp:44-69).  Runs on CPU:
tests/cpu/fem_host_check.cpp is built with hipcc (host code only) against the in-tree library."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dumped(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    lib = os.path.join(ROOT, "palace_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libpalace_amd.so")):
        import __graft_entry__ as ge
        ge.build()
    exe = str(tmp_path_factory.mktemp("fem_host") / "fem_host_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "palace_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpu", "fem_host_check.cpp"),
                           "-L" + lib, "-lpalace_amd", "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, PALACE_AMD_SETUP_THREADS="1"))
    assert out.returncode == 0, out.stdout + out.stderr
    # the host set-up of the coarse hierarchies is row-parallel: whatever the number of threads, every printed value is the same
    out4 = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, PALACE_AMD_SETUP_THREADS="4"))
    assert out4.returncode == 0 and out4.stdout == out.stdout and "amg_threads_checksum" in out.stdout
    vals = {}
    for line in out.stdout.splitlines():
        k, *v = line.split()
        if k == "amg_threads_checksum":  # (compared as text above)
            continue
        if k.startswith("orders") or k.startswith("amg_levels") or k in ("q1d", "mat_dims", "amg_small", "amg_blocks", "amg_thin_blocks"):
            vals[k] = [int(t) for t in v]
        else:
            vals[k] = np.array([struct.unpack("<d", struct.pack("<Q", int(t, 16)))[0] for t in v])
    return vals


def test_point_sets_and_tables(dumped):
    from palace_amd.fem.basis1d import gauss_legendre, gauss_lobatto, lagrange_eval

    for n in range(1, 7):
        x, w = gauss_legendre(n)
        assert np.allclose(dumped[f"gl_x{n}"], x, rtol=0, atol=2e-16) and np.allclose(dumped[f"gl_w{n}"], w, rtol=0, atol=1e-15)
        if n >= 2:
            assert np.allclose(dumped[f"gll{n}"], gauss_lobatto(n), rtol=0, atol=3e-16)
    B, G = lagrange_eval(gauss_lobatto(4), gauss_legendre(4)[0])
    assert np.allclose(dumped["Bc3"], B.reshape(-1), rtol=0, atol=1e-15)
    assert np.allclose(dumped["Gc3"], G.reshape(-1), rtol=0, atol=2e-14)


def test_coefficient_contexts(dumped):
    from palace_amd.ceed import coefficient_context

    def same(name, ref):
        got = dumped[name]
        assert got.size == ref.size, (name, got.size, ref.size)
        assert np.array_equal(got.view(np.uint64) & 0xFFFFFFFF, ref.view(np.uint64) & 0xFFFFFFFF) or np.allclose(got, ref, rtol=1e-15, atol=0), name
        # integer slots carry the int in the low half, real slots must agree to rounding
        assert np.allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=1e-15, atol=1e-300), name

    M = np.array([2.0, 0.1, 0.2, 0.3, 3.0, 0.4, 0.5, 0.6, 4.0]).reshape(3, 3).T  # from column-major
    same("ctx_identity", coefficient_context(3, a=1.5))
    same("ctx_scalar", coefficient_context(3, attr_mat=[0, -1, 0], mat_coeff=[np.array([2.08])]))
    mixed = [2.08 * np.eye(3), 0.5 * M]
    same("ctx_mixed", coefficient_context(3, attr_mat=[0, 1, 0], mat_coeff=mixed))
    same("ctx_mixed_t", coefficient_context(3, attr_mat=[0, 1, 0], mat_coeff=[m.T for m in mixed], a=2.0))
    upd = [(2.08 - 1.0) * np.eye(3), 0.5 * M]
    same("ctx_updated", coefficient_context(3, attr_mat=[0, 1, 0], mat_coeff=upd))
    same("ctx_restricted", coefficient_context(3, attr_mat=[-1, 0, 1], mat_coeff=[upd[1], upd[0]]))
    pair = np.concatenate([coefficient_context(1, attr_mat=[0, 0, 0], mat_coeff=[np.array([0.7])]),
                           coefficient_context(3, attr_mat=[-1, 0, 1], mat_coeff=[upd[1], upd[0]])])
    same("ctx_pair", pair)
    nrm = np.array([0.0, 0.6, 0.8])
    same("ctx_normal", coefficient_context(1, attr_mat=[-1, 0, 1], mat_coeff=[np.array([nrm @ upd[1] @ nrm]), np.array([nrm @ upd[0] @ nrm])]))


def test_coarsening_sequences_and_quadrature(dumped):
    from palace_amd.fem.partition import levels_for

    for p in range(1, 7):
        assert dumped[f"orders_log{p}"] == levels_for(p)
        assert dumped[f"orders_lin{p}"] == list(range(1, p + 1))
    assert dumped["q1d"] == [4]  # order-3 solution: 2p = 6 -> 4 Gauss-Legendre points per direction


def test_matrix_functions(dumped):
    """MatrixSqrt / MatrixPow (linalg/densematrix.cpp:222-252) as the flux estimators use them on symmetric material tensors."""
    S = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.1], [0.0, 0.1, 1.2]])
    w, V = np.linalg.eigh(S)
    assert np.allclose(dumped["mat_sqrt"].reshape(3, 3), (V * np.sqrt(w)) @ V.T, rtol=0, atol=1e-14)
    assert np.allclose(dumped["mat_invsqrt"].reshape(3, 3), (V / np.sqrt(w)) @ V.T, rtol=0, atol=1e-14)
    assert np.allclose(dumped["mat_sqrt"].reshape(3, 3) @ dumped["mat_sqrt"].reshape(3, 3), S, rtol=0, atol=1e-14)
    assert np.allclose(dumped["mat_sqrt_diag"].reshape(3, 3), np.diag([2.0, 3.0, 0.5]), rtol=0, atol=1e-15)
    R = np.array([[3.0, 1.0, 0], [1.0, 3.0, 0], [0, 0, 2.0]])
    assert np.allclose(dumped["mat_square"].reshape(3, 3), R @ R, rtol=0, atol=1e-13)
    # plane problems: 2 x 2 and 1 x 1 MaterialTensors (the matrix functions act on the tensor bordered with an identity block)
    E2 = np.array([[2.0, 0.3], [0.3, 1.5]])
    w, V = np.linalg.eigh(E2)
    got = dumped["mat2_sqrt"].reshape(2, 2, 2)
    assert np.allclose(got[0], (V * np.sqrt(w)) @ V.T, rtol=0, atol=1e-14)
    assert np.allclose(got[1], np.sqrt(3.1) * np.eye(2), rtol=0, atol=1e-14)
    inv = dumped["mat2_invsqrt"].reshape(2, 2, 2)
    assert np.allclose(inv[0] @ got[0], np.eye(2), rtol=0, atol=1e-14)
    assert np.allclose(dumped["mat1_sqrt"], np.sqrt([0.8, 1.4]), rtol=0, atol=1e-15)
    assert dumped["mat_dims"][:2] == [2, 8]


def test_smoothed_aggregation_setup(dumped):
    """The host set-up of the native coarse-level hierarchy (palace_amd/csrc/amg.hpp; stands where the reference calls HYPRE,
    linalg/amg.cpp:12-49): aggregates cover every node once, the tentative prolongator has orthonormal columns, the smoothed one
    keeps constants away from the boundary, the coarse matrix is the Galerkin product, and V-cycles with two Jacobi sweeps
    converge at a grid-independent rate -- also for a strongly anisotropic operator, where the strength filter matters."""
    na, nlev, ncoarse = dumped["amg_small"]
    n = 100
    T = dumped["amg_T"].reshape(n, na)
    assert np.all((T != 0).sum(axis=1) == 1) and np.allclose(T.T @ T, np.eye(na), atol=1e-14)
    assert 3 <= na <= n // 4 and nlev == 2 and ncoarse == na
    P = dumped["amg_P"].reshape(n, na)
    A1 = dumped["amg_A1"].reshape(na, na)
    g = np.arange(10)
    L1 = 2 * np.eye(10) - np.diag(np.ones(9), 1) - np.diag(np.ones(9), -1)
    A0 = np.kron(np.eye(10), L1) + np.kron(L1, np.eye(10))
    assert np.abs(A1 - P.T @ A0 @ P).max() < 1e-13
    assert np.abs(A1 - A1.T).max() < 1e-14 and np.linalg.eigvalsh(A1).min() > 0
    # interior rows: the smoothed prolongator reproduces the (scaled) constant that the tentative one reproduces
    interior = np.array([j * 10 + i for j in range(2, 8) for i in range(2, 8)])
    w = np.linalg.lstsq(T, np.ones(n), rcond=None)[0]
    assert np.abs((P @ w)[interior] - 1.0).max() < 1e-12
    for key, lev in (("amg_factors_iso", "amg_levels0"), ("amg_factors_aniso", "amg_levels1")):
        sizes = dumped[lev]
        assert sizes[0] == 48 * 48 and len(sizes) >= 3 and sizes[-1] <= 60
        assert sizes[1] < 0.5 * sizes[0] and all(b < 0.7 * a for a, b in zip(sizes, sizes[1:]))  # (line aggregates: 1/3 per level)
        f = dumped[key]
        assert f.max() < 0.65 and np.exp(np.log(f[3:]).mean()) < 0.5, (key, f)  # stationary V(2,2), damped Jacobi


def test_hierarchy_of_the_distributed_solve(dumped):
    """amg.hpp: SetupBlocks / amg_dist.hpp: DistSpace (the row-distributed V-cycle that stands where the reference runs HYPRE's
    BoomerAMG on the distributed matrix, linalg/amg.cpp:12-49): aggregates stay inside the ranks' row blocks and are numbered block by
    block; a rank's rows of A_l, R_l, P_l in its local numbering [own | ghosts] reproduce the rows of the global products; and the
    confined hierarchy converges like the unconfined one (stationary V(2,2) with damped Jacobi)."""
    confined, nlev, ncoarse, nlev_global = dumped["amg_blocks"]
    assert confined == 1 and nlev >= 3 and ncoarse <= 60 and abs(nlev - nlev_global) <= 1
    worst, ghosts, plan_mismatch, plan_entries = dumped["amg_blocks_products"]
    assert worst < 1e-13 and ghosts > 0
    # the ranks' exchange plans, each derived on its own: every entry a rank sends lands in the slot of that entry on the receiver
    assert plan_mismatch == 0 and plan_entries > 0
    # thin row blocks whose strong ties all cross the partition: nobody leaves the coarse space (one line per block: singletons;
    # two lines: pairs along the strong direction)
    d1, n1, d2, n2 = dumped["amg_thin_blocks"]
    assert d1 == 0 and d2 == 0 and n1 == 24 * 24 and n2 <= 24 * 24 // 2 + 24, (d1, n1, d2, n2)
    fb, fg = dumped["amg_factors_blocks"], dumped["amg_factors_global"]
    mean = lambda f: np.exp(np.log(f[3:]).mean())  # noqa: E731
    assert fb.max() < 0.65 and mean(fb) < 0.5 and mean(fb) < mean(fg) + 0.1, (fb, fg)


def test_plane_rotations_over_the_whole_exponent_range(tmp_path):
    """krylov_impl.hpp's GeneratePlaneRotation (LAPACK d/zlartg, safe scaling) in tests/cpu/rotation_check.cpp: host code only."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "rotation_check")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "palace_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpu", "rotation_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "worst deviation" in out.stdout, out.stdout + out.stderr
