"""The device cases of the reference's vector unit test (test/unit/test-vector.cpp: "Vector Sum - Real" :16-40,
"Vector Sum - Complex" :75-110, "Sqrt function" :315-348) at communicator size 1, plus sizes that exercise the
16-byte-lane / tail / unaligned paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.fixture(scope="module")
def ctx():
    from palace_amd import linalg

    return linalg.Context()


def test_vector_sum_real_and_complex(ctx):
    rank, size = 0, 1
    v = _dev([rank * 3 + i + 1 for i in range(3)])
    assert ctx.sum(v) == pytest.approx(9.0 * size * (size - 1) / 2.0 + 6.0 * size, rel=1e-15)
    re, im = _dev([rank, rank]), _dev([rank + 0, rank + 1])
    s = complex(ctx.sum(re), ctx.sum(im))  # Sum(ComplexVector) = (Sum(Real), Sum(Imag)), vector.cpp:696-699
    assert s.real == pytest.approx(0.0, abs=1e-15) and s.imag == pytest.approx(1.0, rel=1e-15)


@pytest.mark.parametrize("offset", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 2, 5, 256, 257, 100003])
def test_vector_sum_any_size(ctx, n, offset):
    rng = np.random.default_rng(n + offset)
    a = rng.uniform(-1, 1, n + 2)
    x = _dev(a)[offset:offset + n]
    ref = float(np.sum(a[offset:offset + n]))
    assert abs(ctx.sum(x) - ref) <= 1e-13 * max(1.0, np.abs(a).sum())


def test_sqrt_function(ctx):
    v = _dev([4.0, 9.0, 16.0, 25.0])
    ctx.sqrt(v)
    assert np.array_equal(v.cpu().numpy(), [2.0, 3.0, 4.0, 5.0])
    v = _dev([1.0, 4.0, 9.0])
    ctx.sqrt(v, 4.0)  # sqrt(4 x)
    assert np.array_equal(v.cpu().numpy(), [2.0, 4.0, 6.0])
    rng = np.random.default_rng(0)
    for n, off in ((1, 0), (7, 1), (1000, 0), (1001, 1)):
        a = rng.uniform(0.0, 3.0, n + 2)
        b = _dev(a)
        ctx.sqrt(b[off:off + n], 0.5)
        ref = a.copy()
        ref[off:off + n] = np.sqrt(0.5 * a[off:off + n])
        assert np.abs(b.cpu().numpy() - ref).max() <= 2e-16 * 3.0


def _cv(z):
    import torch

    return (torch.from_numpy(np.ascontiguousarray(z.real)).cuda(), torch.from_numpy(np.ascontiguousarray(z.imag)).cuda())


def _cnp(p):
    return p[0].cpu().numpy() + 1j * p[1].cpu().numpy()


def _cvec_op(ctx, op, coef, x, y=None, z=None, out=None):
    import ctypes as C

    from palace_amd import lib as _lib

    L = _lib.load()
    co = None if coef is None else np.ascontiguousarray(np.array([[c.real, c.imag] for c in map(complex, coef)], dtype=np.float64))
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    ob = (C.c_double * 2)()
    _lib.check(L.pa_cvec_op(ctx.handle, op, x[0].numel(), co.ctypes.data_as(C.c_void_p) if co is not None else None,
                            P(x[0]), P(x[1]), P(y[0] if y else None), P(y[1] if y else None), P(z[0] if z else None),
                            P(z[1] if z else None), ob))
    return complex(ob[0], ob[1])


@pytest.mark.parametrize("n", [1, 255, 1000, 100003])
def test_complex_vector_members(ctx, n):
    """ComplexVector's members beyond Dot / AXPY (linalg/vector.hpp:95-146, vector.cpp:172-460): scaling by a complex
    number, Abs, Reciprocal, Conj, AXPBY, AXPBYPCZ, TransposeDot, SetBlocks -- against numpy complex arithmetic."""
    import ctypes as C

    from palace_amd import lib as _lib

    rng = np.random.default_rng(n)
    cz = lambda: rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)  # noqa: E731
    x, y, z = cz(), cz(), cz()
    a, b, c = 0.7 - 1.3j, -0.4 + 0.2j, 1.1 + 0.6j
    tol = dict(rtol=1e-14, atol=1e-14)
    d = _cv(x); _cvec_op(ctx, 0, [a], d); assert np.allclose(_cnp(d), a * x, **tol)
    d = _cv(x); _cvec_op(ctx, 0, [2.5], d); assert np.allclose(_cnp(d), 2.5 * x, **tol)
    d = _cv(x); _cvec_op(ctx, 1, None, d); assert np.allclose(_cnp(d), np.abs(x), **tol)
    d = _cv(x); _cvec_op(ctx, 2, None, d); assert np.allclose(_cnp(d), 1.0 / x, rtol=1e-13, atol=0)
    d = _cv(x); _cvec_op(ctx, 3, None, d); assert np.array_equal(_cnp(d), np.conj(x))
    dx, dy = _cv(x), _cv(y); _cvec_op(ctx, 4, [a, b], dx, dy); assert np.allclose(_cnp(dy), a * x + b * y, **tol)
    dx, dy, dz = _cv(x), _cv(y), _cv(z)
    _cvec_op(ctx, 5, [a, b, c], dx, dy, dz)
    assert np.allclose(_cnp(dz), a * x + b * y + c * z, **tol)
    td = _cvec_op(ctx, 6, None, _cv(x), _cv(y))
    assert abs(td - np.sum(x * y)) < 1e-12 * max(1.0, np.abs(x * y).sum())
    # SetBlocks
    n1 = n // 3
    blocks = [cz()[:n1], cz()[: n - n1]]
    s = [0.5 + 0.25j, -1.5j]
    dv = [_cv(bk) for bk in blocks]
    out = _cv(np.zeros(n, dtype=complex))
    yr = (C.c_void_p * 2)(*[t[0].data_ptr() for t in dv])
    yi = (C.c_void_p * 2)(*[t[1].data_ptr() for t in dv])
    sizes = (C.c_int * 2)(n1, n - n1)
    sc = np.array([[v.real, v.imag] for v in s], dtype=np.float64)
    if n1 > 0:
        _lib.check(_lib.load().pa_cvec_set_blocks(ctx.handle, C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), n, 2,
                                                  yr, yi, sizes, sc.ctypes.data_as(C.c_void_p)))
        assert np.allclose(_cnp(out), np.concatenate([s[0] * blocks[0], s[1] * blocks[1]]), **tol)


def test_diagonal_operators(ctx):
    """DiagonalOperator / ComplexDiagonalOperator (linalg/operator.hpp:354-423, operator.cpp:415-581): Mult, MultTranspose,
    MultHermitianTranspose and the AddMult forms with a complex coefficient."""
    import ctypes as C
    import torch

    from palace_amd import lib as _lib

    L = _lib.load()
    L.pa_diag_op_apply.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_double, C.c_int, C.c_int]
    n = 4099
    rng = np.random.default_rng(3)
    cz = lambda: rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)  # noqa: E731
    d, x, y0 = cz(), cz(), cz()
    a = 0.3 - 0.8j
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    dd, dx = _cv(d), _cv(x)
    for mode, dm in ((0, d), (1, d), (2, np.conj(d))):
        dy = _cv(y0)
        _lib.check(L.pa_diag_op_apply(ctx.handle, n, P(dd[0]), P(dd[1]), P(dx[0]), P(dx[1]), P(dy[0]), P(dy[1]), 1.0, 0.0, mode, 0))
        assert np.allclose(_cnp(dy), dm * x, rtol=1e-14, atol=1e-15)
        dy = _cv(y0)
        _lib.check(L.pa_diag_op_apply(ctx.handle, n, P(dd[0]), P(dd[1]), P(dx[0]), P(dx[1]), P(dy[0]), P(dy[1]), a.real, a.imag, mode, 1))
        assert np.allclose(_cnp(dy), y0 + a * dm * x, rtol=1e-14, atol=1e-15)
    # real operator
    dr, xr = torch.from_numpy(d.real.copy()).cuda(), torch.from_numpy(x.real.copy()).cuda()
    yr = torch.from_numpy(y0.real.copy()).cuda()
    _lib.check(L.pa_diag_op_apply(ctx.handle, n, P(dr), None, P(xr), None, P(yr), None, 1.0, 0.0, 0, 0))
    assert np.allclose(yr.cpu().numpy(), d.real * x.real, rtol=1e-15, atol=0)
    yr = torch.from_numpy(y0.real.copy()).cuda()
    _lib.check(L.pa_diag_op_apply(ctx.handle, n, P(dr), None, P(xr), None, P(yr), None, -2.5, 0.0, 1, 1))
    assert np.allclose(yr.cpu().numpy(), y0.real - 2.5 * d.real * x.real, rtol=1e-14, atol=1e-15)
