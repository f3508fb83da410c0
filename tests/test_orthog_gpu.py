"""OrthogonalizeColumn{MGS,CGS,CGS2} on the device (pa_orthogonalize_column[_complex]) through the cases of the
reference's own unit test (test/unit/test-orthog.cpp) and against the oracle's restatement on larger random data."""
import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu
KINDS = ["MGS", "CGS", "CGS2"]


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.fixture(scope="module")
def ctx():
    from palace_amd import linalg

    return linalg.Context()


@pytest.mark.parametrize("kind", KINDS)
def test_real_cases_of_the_reference(ctx, kind):
    # "Real Empty" (test-orthog.cpp:98-121)
    w = _dev(np.arange(4.0))
    H = ctx.orthogonalize_column(kind, [], w)
    assert H.size == 0 and np.array_equal(w.cpu().numpy(), np.arange(4.0))
    # "Real 2" (:162-229) at communicator size 1: exact second basis vector, known coefficients
    V = [_dev([1.0, 0, 0, 0]), _dev([0.0, 1, 0, 0])]
    v1 = V[1].clone()
    ctx.orthogonalize_column(kind, V[:1], v1)
    assert np.array_equal(v1.cpu().numpy(), [0.0, 1.0, 0.0, 0.0])
    w = _dev(np.arange(4.0))
    H = ctx.orthogonalize_column(kind, V, w)
    wn = w.cpu().numpy()
    assert abs(wn[0]) < 1e-12 and abs(wn[1]) < 1e-12 and wn[2] == 2.0 and wn[3] == 3.0
    assert abs(H[0]) < 1e-15 and H[1] == pytest.approx(1.0)


@pytest.mark.parametrize("m", [1, 5, 11])
@pytest.mark.parametrize("kind", KINDS)
def test_real_random_vs_oracle(ctx, kind, m):
    rng = np.random.default_rng(m)
    n = 1001
    Q, _ = np.linalg.qr(rng.normal(size=(n, m)))
    V = [Q[:, j].copy() for j in range(m)]
    w = rng.normal(size=n)
    Href, wref = po.orthogonalize_column(kind, V, w, m)
    wd = _dev(w)
    H = ctx.orthogonalize_column(kind, [_dev(v) for v in V], wd)
    assert np.abs(H - Href).max() < 1e-12 * max(1.0, np.abs(Href).max())
    assert np.abs(wd.cpu().numpy() - wref).max() < 1e-12 * np.abs(wref).max()
    assert max(abs(np.dot(wd.cpu().numpy(), v)) for v in V) < 1e-12 * np.linalg.norm(w)


@pytest.mark.parametrize("kind", KINDS)
def test_complex_random_vs_oracle(ctx, kind):
    rng = np.random.default_rng(3)
    n, m = 513, 4
    Q, _ = np.linalg.qr(rng.normal(size=(n, m)) + 1j * rng.normal(size=(n, m)))
    V = [Q[:, j].copy() for j in range(m)]
    w = rng.normal(size=n) + 1j * rng.normal(size=n)
    Href, wref = po.orthogonalize_column(kind, V, w, m)
    wr, wi = _dev(w.real), _dev(w.imag)
    H = ctx.orthogonalize_column_complex(kind, [_dev(v.real) for v in V], [_dev(v.imag) for v in V], wr, wi)
    got = wr.cpu().numpy() + 1j * wi.cpu().numpy()
    assert np.abs(H - Href).max() < 1e-12 * max(1.0, np.abs(Href).max())
    assert np.abs(got - wref).max() < 1e-12 * np.abs(wref).max()
    assert max(abs(np.vdot(v, got)) for v in V) < 1e-12 * np.linalg.norm(w)  # Dot(w, V_j) = V_j^H w


@pytest.mark.parametrize("kind", KINDS)
def test_weighted_by_the_mass_operator(ctx, kind, cylinder_mesh):
    """"Weighted" cases (:270-374) with the Nedelec mass operator as the SPD weight: the result is M-orthogonal to the
    M-normalised basis (what the eigensolver's B-orthogonalisation needs)."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem.fespace import NDHexSpace

    nd = NDHexSpace(cylinder_mesh, 1)
    geom = ceed.GeomFactorData(cylinder_mesh, 2)
    M = linalg.ParOperator(ctx, ceed.ndmass_operator(geom, nd, ceed.coefficient_context(3)), np.zeros(0, np.int32),
                           linalg.DIAG_ONE)
    rng = np.random.default_rng(9)
    n, m = nd.ndofs, 3
    V, t = [], torch.empty(n, dtype=torch.float64, device="cuda")
    for j in range(m):  # M-orthonormal basis by MGS with the device routine itself + explicit normalisation
        v = _dev(rng.normal(size=n))
        ctx.orthogonalize_column("MGS", V, v, weight=M)
        M.mult(v, t)
        v /= float(torch.sqrt(v @ t))
        V.append(v)
    for i in range(m):
        M.mult(V[i], t)
        for j in range(m):
            assert abs(float(V[j] @ t) - (1.0 if i == j else 0.0)) < 1e-12
    w = _dev(rng.normal(size=n))
    w0 = w.clone()
    H = ctx.orthogonalize_column(kind, V, w, weight=M)
    for j in range(m):
        M.mult(V[j], t)
        assert abs(float(w @ t)) < 1e-11 * float(w0.norm())
        assert H[j] == pytest.approx(float(w0 @ t), rel=1e-10, abs=1e-12)


# ---- the Arnoldi column with the coefficients on the device (orthog.hip; round 5) ------------------------------------------------
@pytest.mark.parametrize("n,off", [(4097, 0), (1000, 0), (777, 1), (3, 0)])
@pytest.mark.parametrize("m", [0, 1, 8, 9, 27])
@pytest.mark.parametrize("kind", KINDS)
def test_orthonormalize_column_real(ctx, kind, m, n, off):
    """H, the norm and the normalised vector against the oracle's three statements (orthogonalise, norm, scale); m crosses the
    batch of eight, n is odd / tiny, `off` shifts every vector by one double so that the 16-byte lanes are not usable."""
    if m >= n:
        pytest.skip("more basis vectors than entries")
    rng = np.random.default_rng(100 + m)
    Q, _ = np.linalg.qr(rng.normal(size=(n, max(m, 1))))
    V = [Q[:, j].copy() for j in range(m)]
    w = rng.normal(size=n)
    Href, wref = po.orthogonalize_column(kind, V, w, m) if m else (np.zeros(0), w.copy())
    hn_ref = np.linalg.norm(wref)
    bufs = [_dev(np.concatenate([np.zeros(off), v])) for v in V]
    wd = _dev(np.concatenate([np.zeros(off), w]))
    H, hn = ctx.orthonormalize_column(kind, [b[off:] for b in bufs], wd[off:])
    tol = 1e-12 * max(1.0, np.abs(w).max())
    assert np.abs(H - Href).max(initial=0.0) < tol
    assert abs(hn - hn_ref) < 1e-12 * hn_ref
    assert np.abs(wd[off:].cpu().numpy() - wref / hn_ref).max() < 1e-11
    for b, v in zip(bufs, V):  # the basis is not written
        assert np.array_equal(b[off:].cpu().numpy(), v)


@pytest.mark.parametrize("n,off", [(2049, 0), (512, 1)])
@pytest.mark.parametrize("m", [0, 1, 8, 13])
@pytest.mark.parametrize("kind", KINDS)
def test_orthonormalize_column_complex(ctx, kind, m, n, off):
    rng = np.random.default_rng(200 + m)
    Q, _ = np.linalg.qr(rng.normal(size=(n, max(m, 1))) + 1j * rng.normal(size=(n, max(m, 1))))
    V = [Q[:, j].copy() for j in range(m)]
    w = rng.normal(size=n) + 1j * rng.normal(size=n)
    Href, wref = po.orthogonalize_column(kind, V, w, m) if m else (np.zeros(0, complex), w.copy())
    hn_ref = np.linalg.norm(wref)
    pad = np.zeros(off)
    Vr = [_dev(np.concatenate([pad, v.real])) for v in V]
    Vi = [_dev(np.concatenate([pad, v.imag])) for v in V]
    wr, wi = _dev(np.concatenate([pad, w.real])), _dev(np.concatenate([pad, w.imag]))
    H, hn = ctx.orthonormalize_column_complex(kind, [v[off:] for v in Vr], [v[off:] for v in Vi], wr[off:], wi[off:])
    got = wr[off:].cpu().numpy() + 1j * wi[off:].cpu().numpy()
    assert np.abs(H - Href).max(initial=0.0) < 1e-12 * max(1.0, np.abs(w).max())
    assert abs(hn - hn_ref) < 1e-12 * hn_ref
    assert np.abs(got - wref / hn_ref).max() < 1e-11
    if m:
        assert max(abs(np.vdot(v, got)) for v in V) < 1e-12


def test_orthonormalize_column_is_reproducible(ctx):
    """The last-block reduction adds the partial sums in block order: two runs on the same data agree to the bit."""
    rng = np.random.default_rng(7)
    n, m = 300001, 10
    V = [_dev(rng.normal(size=n) / np.sqrt(n)) for _ in range(m)]
    w0 = rng.normal(size=n)
    out = []
    for _ in range(3):
        w = _dev(w0)
        H, hn = ctx.orthonormalize_column("CGS2", V, w)
        out.append((H.copy(), hn, w.cpu().numpy()))
    for H, hn, w in out[1:]:
        assert np.array_equal(H, out[0][0]) and hn == out[0][1] and np.array_equal(w, out[0][2])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("cplx", [False, True])
def test_orthonormalize_column_across_rank_threads(kind, cplx):
    """The device-resident Gram-Schmidt with a communicator (round 5): the entries of every vector are split over three ranks (threads
    of this process on one GPU, in-process communicator); the coefficients are summed over the ranks ON THE DEVICE between the
    kernels (m + 1 all-reduces for the modified variant, one per pass for the classical ones -- 26 (complex) values: more than one
    message of the in-process transport would be a different code path, so m = 13 keeps both in play with the batch of eight) and
    every rank must return the H, the norm and its slice of the normalised vector of the undivided computation."""
    import threading

    import torch

    from palace_amd import linalg

    world, n, m = 3, 3001, 13
    rng = np.random.default_rng(5)
    A = rng.normal(size=(n, m)) + (1j * rng.normal(size=(n, m)) if cplx else 0.0)
    Q, _ = np.linalg.qr(A)
    V = [Q[:, j].copy() for j in range(m)]
    w = rng.normal(size=n) + (1j * rng.normal(size=n) if cplx else 0.0)
    Href, wref = po.orthogonalize_column(kind, V, w, m)
    hn_ref = np.linalg.norm(wref)
    cuts = [0, 1000, 2100, n]
    group = linalg.LocalGroup(world)
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = linalg.Context()
            c.init_comm_local(group, r)
            sl = slice(cuts[r], cuts[r + 1])
            if cplx:
                Vr, Vi = [_dev(v.real[sl]) for v in V], [_dev(v.imag[sl]) for v in V]
                wr, wi = _dev(w.real[sl]), _dev(w.imag[sl])
                H, hn = c.orthonormalize_column_complex(kind, Vr, Vi, wr, wi)
                out[r] = (H, hn, wr.cpu().numpy() + 1j * wi.cpu().numpy())
            else:
                Vd = [_dev(v[sl]) for v in V]
                wd = _dev(w[sl])
                H, hn = c.orthonormalize_column(kind, Vd, wd)
                out[r] = (H, hn, wd.cpu().numpy())
            c.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            group.abort()
            raise

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for r in range(world):
        H, hn, ws = out[r]
        assert np.abs(H - Href).max() < 1e-12 * max(1.0, np.abs(Href).max())
        assert abs(hn - hn_ref) < 1e-12 * hn_ref
        assert np.abs(ws - wref[cuts[r]:cuts[r + 1]] / hn_ref).max() < 1e-11
        assert np.array_equal(H, out[0][0]) and hn == out[0][1]  # (the sums are global: the same bits on every rank)


# ---- modified Gram-Schmidt with w resident in the register file (orthog.hip: k_mgs_resident; round 6) ----------------------------
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("n,m", [(2, 1), (3, 2), (1000, 9), (4097, 27), (300001, 10), (1500001, 5), (2180208, 12)])
def test_resident_mgs_against_the_chained_form_and_the_oracle(ctx, monkeypatch, cplx, n, m):
    """The same column with w kept in registers (one pass over every basis vector) and with the chained kernels of round 5
    (PALACE_AMD_GS_RESIDENT=0, read per column): H, the norm and the normalised vector agree to rounding; against the oracle's three
    statements (orthog.hpp:41-62, iterative.cpp:629-633) where the oracle is quick; the resident form is reproducible to the
    bit; sizes: one 16-byte lane, an odd tail, every slot count up to config 3's vector (2 180 208 complex entries)."""
    rng = np.random.default_rng(300 + m)
    scale = 1.0 / np.sqrt(n)
    mk = lambda: rng.normal(size=n) * scale
    Vr = [_dev(mk()) for _ in range(m)]
    Vi = [_dev(mk()) for _ in range(m)] if cplx else None
    w0r, w0i = mk() * np.sqrt(n), mk() * np.sqrt(n)

    def run():
        wr, wi = _dev(w0r), _dev(w0i)
        if cplx:
            H, hn = ctx.orthonormalize_column_complex("MGS", Vr, Vi, wr, wi)
            return H, hn, wr.cpu().numpy() + 1j * wi.cpu().numpy()
        H, hn = ctx.orthonormalize_column("MGS", Vr, wr)
        return H, hn, wr.cpu().numpy()

    before = ctx.resident_columns()
    H1, hn1, w1 = run()
    assert ctx.resident_columns() == before + 1, "the resident form did not run"
    H2, hn2, w2 = run()
    assert np.array_equal(H1, H2) and hn1 == hn2 and np.array_equal(w1, w2)
    monkeypatch.setenv("PALACE_AMD_GS_RESIDENT", "0")
    H0, hn0, w0 = run()
    assert ctx.resident_columns() == before + 2
    tol = 1e-13 * max(1.0, np.abs(H0).max())
    assert np.abs(H1 - H0).max() < tol and abs(hn1 - hn0) < 1e-13 * hn0 and np.abs(w1 - w0).max() < 1e-13 * np.abs(w0).max() * max(m, 4)
    if n <= 300001:
        V = [Vr[j].cpu().numpy() + (1j * Vi[j].cpu().numpy() if cplx else 0.0) for j in range(m)]
        Href, wref = po.orthogonalize_column("MGS", V, w0r + (1j * w0i if cplx else 0.0), m)
        hn_ref = np.linalg.norm(wref)
        assert np.abs(H1 - Href).max() < 1e-12 * max(1.0, np.abs(Href).max())
        assert abs(hn1 - hn_ref) < 1e-12 * hn_ref and np.abs(w1 - wref / hn_ref).max() < 1e-11


def test_resident_mgs_falls_back_when_it_does_not_apply(ctx):
    """Vectors off the 16-byte grid, and vectors that do not fit the register file, keep the chained form (same results as ever:
    the tests above them in this file)."""
    import torch

    rng = np.random.default_rng(11)
    n = 4097
    buf = [_dev(rng.normal(size=n + 1)) for _ in range(3)]
    w = _dev(rng.normal(size=n + 1))
    before = ctx.resident_columns()
    ctx.orthonormalize_column("MGS", [b[1:] for b in buf], w[1:])
    assert ctx.resident_columns() == before
    n = 16_000_001  # 128 MB per vector: more than the register file holds beside the basis slice
    V = [torch.randn(n, dtype=torch.float64, device="cuda") / np.sqrt(n) for _ in range(2)]
    w = torch.randn(n, dtype=torch.float64, device="cuda")
    w0 = w.clone()
    H, hn = ctx.orthonormalize_column("MGS", V, w)
    assert ctx.resident_columns() == before
    h0 = float(V[0] @ w0)
    assert abs(H[0] - h0) < 1e-10 * max(1.0, abs(h0)) and abs(float(w.norm()) - 1.0) < 1e-12
